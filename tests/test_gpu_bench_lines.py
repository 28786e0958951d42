"""GPU: bench.py prints ONE well-formed JSON line for the headline mode and for BASELINE config #4 (--auto-randoms), on sizes
that take seconds: the fields the driver and the judge read are there and consistent (counter-based fields may be null -- the
committed counters belong to the full-size workloads)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(*args):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, cwd=ROOT,
                         timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def test_headline_mode_line():
    d = _bench("--n-ref", "3e5", "--n-unk", "3e5", "--patches", "16", "--steps", "3", "--warmup", "1", "--cpu-seconds", "2")
    assert d["metric"] == "candidate pairs/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "fixed_cost_ms", "fp32_frac", "valu_issue_frac"):
        assert key in r, key
    assert r["bound"] == "valu_fp64" and 0 < r["frac"] < 2 and r["fp32_frac"] > 0
    assert r["launch_ms"] < d["ms_per_step"]
    assert r["exact_sample"]["parity_with_default_path"] is True
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["parity_with_gpu"] is True and "sample" in c
    assert "multi_gpu" not in d


def test_config4_mode_line():
    d = _bench("--n-ref", "2e5", "--auto-randoms", "6e5", "--weights", "--patches", "16", "--steps", "3", "--warmup", "1",
               "--cpu-seconds", "1")
    assert d["metric"] == "candidate pairs/s" and d["value"] > 0 and "autocorrelate" in d["config"]["workload"]
    counts = d["roofline"]["counts"]
    assert set(counts) == {"DD", "DR", "RR"}
    for k, c in counts.items():
        assert c["count_kernel_ms"] > 0 and c["evaluated_entries"] > 0 and c["candidate_pairs"] > 0, k
    assert d["candidate_pairs_per_step"] == sum(c["candidate_pairs"] for c in counts.values())
    assert d["ms_per_step"] >= d["roofline"]["count_kernels_ms"]
    assert d["cpu_baseline"]["parity_with_gpu"] is True
