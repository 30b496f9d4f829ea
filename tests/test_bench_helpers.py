"""CPU: the pure pieces of bench.py's line -- how the host-fed legs of several ranks are folded, what the roofline block says for a
workload with an end-to-end byte count, and which profile constants a config may use."""
import json
import os
import types

import bench


def test_gather_extras_sums_numeric_leaves_over_ranks():
    class FakeDist:
        def __init__(self, objs):
            self.objs = objs

        def all_gather_object(self, out, local):
            for i, o in enumerate(self.objs):
                out[i] = o

    ranks = [{"host_fed_cuts_per_s": {"batch_60": 100.0 + r, "batch_1024": 200.0, "what": "text"}, "flag": True} for r in range(4)]
    got = bench.gather_extras(ranks[0], FakeDist(ranks), 4)
    assert got["aggregate_over_ranks"]["host_fed_cuts_per_s"] == {"batch_60": 406.0, "batch_1024": 800.0}  # strings and bools are not summed
    assert got["per_rank"] == ranks and "CONCURRENTLY" in got["what"]
    assert bench.gather_extras(ranks[0], None, 1) is ranks[0]


def test_roofline_block_prices_the_on_the_fly_step_both_ways(monkeypatch):
    monkeypatch.setattr(bench, "load_profile_constants", lambda kernel, name="fbank16k": {"hbm_bytes_per_algorithmic_byte": 1.25, "per_kernel": {"k": 1}} if name == "onthefly" else {})
    w = types.SimpleNamespace(algo_bytes=8_000_000_000, algo_bytes_end_to_end=4_000_000_000, kernel="fft512c_kernel<13> x", units=100,
                              algo_parts={"feature_launches_per_step": 64})
    r, prof = bench.roofline_block(w, 2.0, "onthefly")
    assert r["frac"] == 0.5 and r["frac_end_to_end"] == 0.25 and r["traffic"] == 10_000_000_000 and r["traffic_per_kernel"] == {"k": 1}
    assert "frac_note" in r and r["algorithmic_bytes_parts"]["feature_launches_per_step"] == 64
    w2 = types.SimpleNamespace(algo_bytes=9_600_000_000, kernel="other", units=10000)
    r2, _ = bench.roofline_block(w2, 3.0, "fbank16k")
    assert r2["traffic"] is None and "frac_end_to_end" not in r2 and r2["frac"] == 0.4


def test_committed_traffic_json_serves_all_three_baseline_kernels_while_the_sources_match():
    with open(os.path.join(bench.ROOT, "profiles", "traffic.json")) as f:
        t = json.load(f)
    assert set(t["configs"]) == {"mfcc40_libri", "onthefly"}
    head = bench.load_profile_constants("fft512c_kernel<13> fbank lds=81152B")
    assert head.get("stale") or head["hbm_bytes_per_cut"] > 960000
    for name in ("mfcc40_libri", "onthefly"):
        c = bench.load_profile_constants("fft512c_kernel<13> whatever", name)
        assert c.get("stale") or 1.0 <= c["hbm_bytes_per_algorithmic_byte"] < 1.5
    assert bench.load_profile_constants("some_other_kernel", "onthefly") == {}
