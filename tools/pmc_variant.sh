# PMC counters of the count kernel for one compile-time variant:  tools/pmc_variant.sh "<flags>" <tag>
set -e
cd ${GRAFT_REPO_ROOT:-.}
python -c "
from yet_another_wizz_amd import build
build.build_library(force=True, extra_flags='$1'.split())" 2>/dev/null
export TMPDIR=/tmp; O=$PWD/gpurun_out/pmc_$2; mkdir -p $O; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $O -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-seconds 0 > /dev/null 2>$O/err.txt
python3 - <<PY
import csv
seen={}
for r in csv.DictReader(open("$O/run_counter_collection.csv")):
    if "k_count_merged" in r["Kernel_Name"]:
        seen.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        dur = int(r["End_Timestamp"])-int(r["Start_Timestamp"])
print("$2", "dur_ms", dur/1e6, {k: f"{v[-1]:.3g}" for k,v in sorted(seen.items())})
PY
