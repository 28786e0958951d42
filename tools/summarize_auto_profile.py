"""Condense tools/profile_cmd.sh output of tools/probe_auto_prof.py into profiles/<name>_{DD,DR,RR}_{pmc,sq_counters}.json.

    python tools/summarize_auto_profile.py gpurun_out/prof_<tag> <name> [reps]

The probe launches the count kernel REPS times per count, DD then DR then RR, and nothing else that matches ``k_count``:
the k-th group of REPS count-kernel dispatches of every pass belongs to count k. Values are those of the LAST dispatch of a
group (warm). HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 as MI355X_MICROARCH.md prescribes for gfx950."""
import csv
import glob
import json
import os
import re
import shutil
import sys

src, name = sys.argv[1:3]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4  # 0: the counts CYCLE (DD, DR, RR, DD, ...), as under bench.py --auto-randoms
labels = sys.argv[4].split(",") if len(sys.argv) > 4 and sys.argv[4] else ["DD", "DR", "RR"]
traffic_prefix = sys.argv[5] if len(sys.argv) > 5 else None  # e.g. "autocorr:{}:10000000+100000000:p64:b30:w1" -> profiles/pmc_traffic.json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
sys.path.insert(0, root)
from yet_another_wizz_amd.build import source_sha16  # noqa: E402

sha = source_sha16()


def groups(sub):
    """{label: {counter: value, 'ns': duration, 'kernel': name, ...}} from one pass."""
    files = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)
    res = {}
    if not files:
        return res
    rows = [r for r in csv.DictReader(open(files[0])) if "k_count" in r["Kernel_Name"]]
    by_dispatch = {}
    for r in rows:
        by_dispatch.setdefault(int(r["Dispatch_Id"]), []).append(r)
    for i, did in enumerate(sorted(by_dispatch)):
        if reps == 0:
            lab = labels[i % len(labels)]
        else:
            lab = labels[i // reps] if i // reps < len(labels) else f"extra{i // reps}"
        rec = {}
        for r in by_dispatch[did]:
            rec[r["Counter_Name"]] = float(r["Counter_Value"])
            rec["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            m = re.search(r"k_count\w*(<[^>]*>)?", r["Kernel_Name"])
            rec["kernel"] = m.group(0) if m else r["Kernel_Name"][:80]
            rec["grid"], rec["vgpr"], rec["sgpr"], rec["lds"] = int(r["Grid_Size"]), int(r["VGPR_Count"]), int(r["SGPR_Count"]), int(r["LDS_Block_Size"])
        res[lab] = rec  # the last dispatch of the group wins
    return res


fetch, write = groups("pmc_fetch"), groups("pmc_write")
sq = [groups(s) for s in ("sq1", "sq2", "sq3")]
stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(out, f"{name}_kernel_stats.csv"))
log = os.path.join(src, "stats.log")
if os.path.exists(log):
    shutil.copy(log, os.path.join(out, f"{name}_under_rocprof.log"))
for lab in labels:
    if lab in fetch and lab in write:
        f, w = fetch[lab], write[lab]
        nbytes = (2.0 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024.0
        rec = dict(kernel=f["kernel"], grid=f["grid"], vgpr=f["vgpr"], sgpr=f["sgpr"], lds=f["lds"], FETCH_SIZE=f["FETCH_SIZE"],
                   WRITE_SIZE=w["WRITE_SIZE"], hbm_bytes=nbytes, kernel_ms_under_pmc=f["ns"] / 1e6,
                   achieved_hbm_gbps=nbytes / f["ns"], source_sha16=sha,
                   method="(2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, rocprofv3 --pmc, one counter per pass, last of %d launches" % reps)
        json.dump(rec, open(os.path.join(out, f"{name}_{lab}_pmc.json"), "w"), indent=1)
        if traffic_prefix:
            import datetime
            import subprocess

            m = re.search(r"k_count_band32(_fine)?", f["kernel"])
            variant = 33 if (m and m.group(1)) else (32 if m else 64)
            tfile = os.path.join(out, "pmc_traffic.json")
            table = json.load(open(tfile)) if os.path.exists(tfile) else {}
            commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
            s1 = (sq[0].get(lab) or {}) if sq else {}
            table[traffic_prefix.format(lab) + f":v{variant}"] = dict(
                bytes=nbytes, source=f"profiles/{name}_{lab}_pmc.json", commit=commit or None, source_sha16=sha,
                sq_insts_valu=s1.get("SQ_INSTS_VALU"), sq_active_inst_valu=s1.get("SQ_ACTIVE_INST_VALU"),
                sq_source=f"profiles/{name}_{lab}_sq_counters.json", count_kernel_ns=f["ns"], date=datetime.date.today().isoformat(),
                method="(2*FETCH_SIZE + WRITE_SIZE)*1024 per count-kernel launch, rocprofv3 --pmc, one counter per pass")
            json.dump(table, open(tfile, "w"), indent=1)
        print(lab, rec["kernel"], f"{nbytes / 1e9:.3f} GB  {f['ns'] / 1e6:.3f} ms  {nbytes / f['ns']:.0f} GB/s")
    rec = {}
    for i, g in enumerate(sq):
        if lab in g:
            d = dict(g[lab])
            d["kernel_ms_under_pmc"] = d.pop("ns") / 1e6
            rec[f"sq{i + 1}"] = d
    if rec:
        rec["_about"] = ("SQ counters of the count kernel, one launch, rocprofv3 --pmc passes (tools/profile_cmd.sh); "
                         "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)")
        rec["source_sha16"] = sha
        json.dump(rec, open(os.path.join(out, f"{name}_{lab}_sq_counters.json"), "w"), indent=1)
        s1 = rec.get("sq1", {})
        if s1:
            wc = s1.get("SQ_WAVE_CYCLES", 0) or 1
            print(lab, "wave-cycle shares: wait_any %.2f wait_inst %.2f active %.2f (valu %.2f lds %.2f)" % (
                s1.get("SQ_WAIT_ANY", 0) / wc, s1.get("SQ_WAIT_INST_ANY", 0) / wc, s1.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                s1.get("SQ_ACTIVE_INST_VALU", 0) / wc, s1.get("SQ_ACTIVE_INST_LDS", 0) / wc))
