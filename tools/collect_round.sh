#!/bin/bash
# Bench lines and probes of one round on ONE box -> gpurun_out/round_<tag>/ (copy what is to be judged into profiles/).
#   tools/collect_round.sh r03
TAG=${1:-rXX}
O=gpurun_out/round_$TAG; mkdir -p $O
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/${TAG}_${name}_bench.json 2> $O/${name}.err; python tools/bench_line.py "$name" < $O/${TAG}_${name}_bench.json; }
run band_10Mx10M --steps 20 --warmup 5
run band_10Mx10M_weighted --steps 20 --warmup 5 --weights --cpu-seconds 0
run band_10Mx10M_kpc --steps 20 --warmup 5 --kpc --cpu-seconds 0
run band_10Mx10M_rweight --steps 20 --warmup 5 --rweight -1 --cpu-seconds 0
run band_1Mx1M --steps 50 --warmup 10 --n-ref 1e6 --n-unk 1e6 --patches 16 --cpu-seconds 0
run band_25Mx25M --steps 10 --warmup 3 --n-ref 2.5e7 --n-unk 2.5e7 --cpu-seconds 0
run band_50Mx50M_3scales --steps 5 --warmup 2 --n-ref 5e7 --n-unk 5e7 --patches 128 --scales 3 --cpu-seconds 0
run band_100Mx100M --steps 5 --warmup 2 --n-ref 1e8 --n-unk 1e8 --patches 128 --cpu-seconds 0
run band64_10Mx10M --steps 20 --warmup 5 --set band_fp32=0 --cpu-seconds 0
timeout -k 10 400 python tools/probe_auto.py 1e7 1e8 w > $O/${TAG}_autocorr_10M_100M_weighted.log 2>&1; grep -E "auto->|end to end|kernels:" $O/${TAG}_autocorr_10M_100M_weighted.log
YAW_KERNELS="auto" timeout -k 10 400 python tools/probe_clustered.py > $O/${TAG}_clustered_3Mx4M.log 2>&1; tail -4 $O/${TAG}_clustered_3Mx4M.log
