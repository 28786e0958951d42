"""bench.py bookkeeping that needs no GPU: a bench line may only quote counters (profiles/pmc_traffic.json) taken on its own
workload AND its own kernel variant, and only while the kernel sources still hash to what they were taken with."""
import json
import os
import types

import bench
from conftest import ROOT


def _args(**kw):
    base = dict(n_ref=10e6, n_unk=10e6, patches=64, zbins=30, scales=1, weights=False, rweight=None, kpc=False)
    base.update(kw)
    return types.SimpleNamespace(**base)


def test_traffic_keys_name_workload_and_kernel_variant():
    keys = {
        bench.traffic_key(_args(), "band", 32),
        bench.traffic_key(_args(weights=True), "band", 32),
        bench.traffic_key(_args(rweight=-1.0), "band", 33),
        bench.traffic_key(_args(kpc=True), "band", 32),
        bench.traffic_key(_args(), "band", 64),            # band_fp32 = 0: another kernel, other counters
        bench.traffic_key(_args(scales=3, n_ref=50e6, n_unk=50e6, patches=128), "band", 32),
        bench.traffic_key(_args(), "exact", None),
    }
    assert len(keys) == 7
    assert bench.traffic_key(_args(rweight=-1.0), "band", 33).endswith(":rw50:v33")
    assert ":kpc:" in bench.traffic_key(_args(kpc=True), "band", 32)


def test_committed_counters_are_keyed_as_bench_asks_for_them():
    """Every band entry of the committed table carries the variant suffix, its provenance and one source hash."""
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
        table = json.load(f)
    entries = {k: v for k, v in table.items() if isinstance(v, dict) and (k.startswith("band:") or k.startswith("autocorr:"))}
    assert bench.traffic_key(_args(), "band", 32) in entries  # the headline
    assert bench.traffic_key(_args(weights=True), "band", 32) in entries
    assert bench.traffic_key(_args(kpc=True), "band", 32) in entries
    assert bench.traffic_key(_args(rweight=-1.0), "band", 33) in entries
    assert any(k.startswith("autocorr:RR:") for k in entries)
    hashes = set()
    for key, e in entries.items():
        assert key.rsplit(":", 1)[1] in ("v32", "v33", "v64"), key
        assert e["bytes"] > 0 and e["source"].startswith("profiles/") and os.path.exists(os.path.join(ROOT, e["source"])), key
        assert e.get("sq_active_inst_valu") and e.get("sq_insts_valu"), key
        hashes.add(e["source_sha16"])
    assert len(hashes) == 1  # one collection run of one build


def test_reference_timing_finds_the_committed_runs():
    assert bench.reference_timing(_args())["seconds"] > 5
    assert bench.reference_timing(_args(weights=True))["seconds"] > 5       # round 3 returned nothing for --weights
    assert bench.reference_timing(_args(kpc=True)) is None                  # not what the reference was timed on
    assert bench.reference_timing(_args(n_ref=3e6)) is None
