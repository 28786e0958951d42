"""Condense bench.py's JSON line (stdin) to the few numbers an A/B run compares:  python bench.py ... | python tools/bench_line.py <tag>"""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
for raw in sys.stdin:
    raw = raw.strip()
    if not raw.startswith("{"):
        continue
    d = json.loads(raw)
    r = d["roofline"]
    print(tag, "ms/step", round(d["ms_per_step"], 3), "kernels_ms", round(d["kernel_ms_per_step"], 3), "count_ms",
          round(r["launch_ms"], 3), "frac", round(r["frac"], 3), "evaluated %.3e" % d["evaluated_pairs_per_step"], flush=True)
