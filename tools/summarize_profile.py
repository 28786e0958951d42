"""Turn the output of tools/profile_bench.sh into the files kept under profiles/.

    python tools/summarize_profile.py gpurun_out/prof_<tag> <name> <traffic key>

writes profiles/<name>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary),
profiles/<name>_bench_under_rocprof.json (bench line of the same run), profiles/<name>_pmc.json
(per-launch FETCH_SIZE / WRITE_SIZE of the item builder and the count kernel) and updates
profiles/pmc_traffic.json[<traffic key>] = HBM bytes per count-kernel launch, computed as
MI355X_MICROARCH.md prescribes for gfx950: (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE counts
64-byte... units of kB on this part are half of the true fetch; separate passes per counter)."""
import csv
import glob
import json
import os
import shutil
import sys

src, name, key = sys.argv[1:4]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)
stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(out, f"{name}_kernel_stats.csv"))
bench = os.path.join(src, "bench_stats.json")
if os.path.exists(bench):
    shutil.copy(bench, os.path.join(out, f"{name}_bench_under_rocprof.json"))
pmc = {}
for counter, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    files = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            kn = row["Kernel_Name"]
            kind = "build" if "k_build_items" in kn else "count" if "k_count" in kn else None
            if kind is None:
                continue
            pmc.setdefault(f"{kind}:{counter}", []).append(dict(
                value=float(row["Counter_Value"]), grid=int(row["Grid_Size"]), wg=int(row["Workgroup_Size"]),
                vgpr=int(row["VGPR_Count"]), sgpr=int(row["SGPR_Count"]), lds=int(row["LDS_Block_Size"]),
                ns=int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
with open(os.path.join(out, f"{name}_pmc.json"), "w") as f:
    json.dump(pmc, f, indent=1)
sys.path.insert(0, root)
from yet_another_wizz_amd.build import source_sha16  # noqa: E402

sha = source_sha16()
# SQ counters of the count kernel (tools/profile_bench.sh passes sq1, sq2): last launch of every counter
sq = {}
for sub in ("sq1", "sq2"):
    files = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    seen, dur, kname = {}, 0, None
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            if "k_count" in row["Kernel_Name"]:
                seen[row["Counter_Name"]] = float(row["Counter_Value"])
                dur = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                import re

                m = re.search(r"k_count\w*(<[^>]*>)?", row["Kernel_Name"])
                kname = m.group(0) if m else row["Kernel_Name"][:80]
    if seen:
        sq[sub] = dict(kernel_ms_under_pmc=dur / 1e6, kernel=kname, **seen)
if sq:
    sq["_about"] = ("SQ counters of the count kernel, one launch, two rocprofv3 --pmc passes (tools/profile_bench.sh); "
                    "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)")
    sq["source_sha16"] = sha
    with open(os.path.join(out, f"{name}_sq_counters.json"), "w") as f:
        json.dump(sq, f, indent=1)
mean = lambda rows: sum(r["value"] for r in rows) / max(len(rows), 1)
if "count:FETCH_SIZE" in pmc and "count:WRITE_SIZE" in pmc:
    traffic = (2.0 * mean(pmc["count:FETCH_SIZE"]) + mean(pmc["count:WRITE_SIZE"])) * 1024.0
    tfile = os.path.join(out, "pmc_traffic.json")
    table = json.load(open(tfile)) if os.path.exists(tfile) else {}
    import datetime
    import subprocess

    commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    table[key] = dict(bytes=traffic, source=f"profiles/{name}_pmc.json", commit=commit or None, source_sha16=sha,
                      sq_insts_valu=(sq.get("sq1") or {}).get("SQ_INSTS_VALU"),
                      sq_active_inst_valu=(sq.get("sq1") or {}).get("SQ_ACTIVE_INST_VALU"), sq_source=f"profiles/{name}_sq_counters.json" if sq else None,
                      count_kernel_ns=mean([dict(value=r["ns"]) for r in pmc["count:FETCH_SIZE"]]),
                      date=datetime.date.today().isoformat(),
                      method="(2*FETCH_SIZE + WRITE_SIZE)*1024 per count-kernel launch, rocprofv3 --pmc, one counter per pass")
    with open(tfile, "w") as f:
        json.dump(table, f, indent=1)
    print(f"{key}: {traffic:.4g} HBM bytes per count-kernel launch "
          f"(fetch {mean(pmc['count:FETCH_SIZE']):.1f} kB-units, write {mean(pmc['count:WRITE_SIZE']):.1f})")
