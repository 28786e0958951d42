# SQ counters of the count kernel, two passes of 8 counters:  tools/pmc_counters.sh <tag> [bench args...]
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
export TMPDIR=/tmp; O=$R/gpurun_out/pmc_$TAG; mkdir -p $O; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/p1 -o run -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 "$@" > /dev/null 2>$O/err1.txt
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/p2 -o run -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 "$@" > /dev/null 2>$O/err2.txt
python3 - <<PY
import csv, glob
for p in ("p1", "p2"):
    seen = {}
    dur = 0
    for f in glob.glob("$O/%s/*counter_collection.csv" % p):
        for r in csv.DictReader(open(f)):
            if "k_count_merged" in r["Kernel_Name"] or "k_count_band" in r["Kernel_Name"]:
                seen.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print("$TAG", p, "dur_ms", dur / 1e6, {k: f"{v[-1]:.4g}" for k, v in sorted(seen.items())})
PY
