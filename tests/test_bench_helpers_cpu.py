"""bench.py's pure helpers (no GPU): the stated N-GPU prediction, the HBM-honest block, and the staleness rules of
profiles/traffic.json (kernel name AND launch geometry)."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_prediction_model():
    b = _bench()
    one = b.predicted_step("cfg3", 1, 2048 * 2048 * 4)
    assert one["predicted_speedup"] == 1.0 and one["predicted_allreduce_ms"] == 0.0
    eight = b.predicted_step("cfg3", 8, 2048 * 2048 * 4)
    # (1680 - 1.5) / 8 + 1.5 + 2 * 7/8 * 16.8 MB / 100 GB/s
    assert abs(eight["predicted_step_ms"] - ((1680.0 - 1.5) / 8 + 1.5 + 2 * 7 / 8 * 2048 * 2048 * 4 / 100e9 * 1e3)) < 1e-9
    assert 7.5 < eight["predicted_speedup"] < 8.0
    assert 7.8 < b.predicted_step("cfg4", 8, 4096 * 4096 * 4)["predicted_speedup"] < 8.0
    assert b.predicted_step("odd2000", 8, 1) is None
    for w in ("cfg1", "cfg2", "cfg3", "cfg4", "cfg5"):
        assert w in b.PREDICTION and w in b.WORKLOADS


def test_hbm_honest_block():
    b = _bench()
    kern = {"xpass": {"traffic": 60 * 72.8e6, "items_per_launch": 60.0, "avg_launch_ms": 0.817, "fabric_frac": 0.65},
            "ypass": {"traffic": 60 * 73.9e6, "items_per_launch": 60.0, "avg_launch_ms": 1.307, "fabric_frac": 0.41}}
    plan4 = {"box_rows": 2049, "batch": 60, "planes_in_flight": 1}
    h = b.hbm_streaming(kern, plan4, 4096)
    assert h["hbm_streaming"] is True and abs(h["bytes_per_item"] - 146.7e6) < 1e3
    assert abs(h["hbm_GBs"] - 146.7e6 / ((0.817 + 1.307) / 60 * 1e-3) / 1e9) < 1e-6 and 0.45 < h["hbm_frac"] < 0.55
    # config 3: twelve 16.8 MB items stay inside the 256 MiB cache -- no HBM claim
    h3 = b.hbm_streaming(kern, {"box_rows": 1025, "batch": 12, "planes_in_flight": 1}, 2048)
    assert h3["hbm_streaming"] is False and "hbm_frac" not in h3
    # streaming, but no current counters
    h0 = b.hbm_streaming({"xpass": {}, "ypass": {}}, plan4, 4096)
    assert h0["hbm_streaming"] is True and "hbm_frac" not in h0


def test_traffic_entries_are_dropped_on_kernel_or_geometry_mismatch():
    b = _bench()
    entry = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["cfg4"]
    geo = entry["geometry"]
    assert set(geo) == set(b.GEOMETRY_KEYS)

    def kern():
        return {"xpass": {"kernel": entry["xpass_kernel"], "avg_launch_ms": 0.8, "items_per_launch": 60.0},
                "ypass": {"kernel": entry["ypass_kernel"], "avg_launch_ms": 1.3, "items_per_launch": 60.0}}
    k = kern()
    src = b.attach_traffic("cfg4", k, dict(geo))
    assert src and "traffic" in k["xpass"] and "traffic" in k["ypass"] and "traffic_stale" not in k["ypass"]
    assert abs(k["ypass"]["traffic"] - 60 * entry["ypass_bytes_per_item"]) < 1
    k = kern()
    b.attach_traffic("cfg4", k, dict(geo, batch=geo["batch"] // 2))            # a host-side change: same kernels, other batch
    assert "traffic" not in k["xpass"] and "geometry" in k["xpass"]["traffic_stale"]
    k = kern()
    k["ypass"]["kernel"] = "k_ypass_coop<12, 4>"                               # another kernel
    b.attach_traffic("cfg4", k, dict(geo))
    assert "traffic" in k["xpass"] and "traffic" not in k["ypass"] and "k_ypass_coop<12, 4>" in k["ypass"]["traffic_stale"]
    assert b.attach_traffic("no-such-workload", kern(), dict(geo)) is None or True
