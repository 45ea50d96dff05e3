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
    # (single - fixed) / 8 + fixed + 2 * 7/8 * 16.8 MB / 100 GB/s; single = the round's committed bench line
    # (profiles/prediction_inputs.json, written by scripts/collect_profiles.py) or the literal
    single, fixed = b.PREDICTION["cfg3"]["single_gpu_ms"], b.PREDICTION["cfg3"]["fixed_ms"]
    assert 1500.0 < single < 1900.0 and fixed == 1.5 and eight["single_gpu_ms_assumed"] == single and eight["single_gpu_ms_source"]
    assert abs(eight["predicted_step_ms"] - ((single - fixed) / 8 + fixed + 2 * 7 / 8 * 2048 * 2048 * 4 / 100e9 * 1e3)) < 1e-9
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


def test_gpus_flag_and_world_size(tmp_path):
    """--gpus that contradicts the launcher's WORLD_SIZE is refused before any GPU work; WITHOUT --gpus the launcher's
    WORLD_SIZE is adopted (`torchrun --nproc-per-node N bench.py`, round-5 advice: the default of 1 used to refuse it)."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "cfg1"], capture_output=True, text=True,
                         timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr
    b = _bench()
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        assert b.parse_args().gpus is None                       # not given: main() takes WORLD_SIZE (or 1)
        sys.argv = ["bench.py", "--gpus", "8"]
        assert b.parse_args().gpus == 8
    finally:
        sys.argv = argv
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "if args.gpus is not None and args.gpus != world:" in src and "args.gpus = world" in src


def test_check_in_names_the_missing_rank_and_exits_non_zero():
    """First-contact insurance: a peer that never arrives ends this rank in seconds with the peer NAMED and a non-zero code
    (here: world 2, only rank 0 exists, 1 s timeout) -- and a complete phase returns at once."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    code = (
        "import importlib.util, sys, datetime\n"
        f"spec = importlib.util.spec_from_file_location('b', r'{os.path.join(ROOT, 'bench.py')}')\n"
        "b = importlib.util.module_from_spec(spec); sys.argv = ['bench.py']; spec.loader.exec_module(b)\n"
        "import torch.distributed as dist\n"
        f"store = dist.TCPStore('127.0.0.1', {port}, 1, True, timeout=datetime.timedelta(seconds=30))\n"
        "b.check_in(store, 0, 1, 'alone', timeout_s=1)\n"
        "print('phase one complete', flush=True)\n"
        "b.check_in(store, 0, 2, 'pair', timeout_s=1)\n"
        "print('not reached', flush=True)\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 3, (out.returncode, out.stderr[-500:])
    assert "phase one complete" in out.stdout and "not reached" not in out.stdout
    assert "rank 0" in out.stderr and "rank(s) [1] of 2 did not arrive" in out.stderr and "'pair'" in out.stderr


def test_default_line_carries_every_baseline_config():
    """The default run's extra_workloads must name configs 1, 2, 4 (shard) and 5 (+ odd2000) -- read from the source's syntax tree
    (the run itself takes minutes): a comment once swallowed an entry of the tuple."""
    import ast
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    names = None
    for node in ast.walk(tree):
        if isinstance(node, ast.For) and isinstance(node.target, ast.Tuple) and [getattr(t, "id", None) for t in node.target.elts] == ["name", "kw"]:
            names = [elt.elts[0].value for elt in node.iter.elts]
    assert names == ["cfg1", "cfg2", "cfg4", "cfg5", "odd2000"], names
