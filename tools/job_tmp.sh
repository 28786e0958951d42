YAWHIP_TRACE=1 YAW_FULL=0 YAW_KERNELS=auto timeout -k 10 300 python tools/probe_clustered.py 2>&1 | grep -E "cross auto|auto auto|run skew" | sort | uniq -c | sort -rn | head -8
YAWHIP_TRACE=1 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 2>&1 | grep -E "run skew" | sort | uniq -c
YAWHIP_TRACE=1 python tools/probe_auto_prof.py 1e7 1e8 w 2>&1 | grep -E "run skew|^DD|^DR|^RR" | sort | uniq -c | cut -c1-150
YAWHIP_TRACE=1 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --n-ref 1e6 --n-unk 1e6 --patches 16 2>&1 | grep -E "run skew" | sort | uniq -c
