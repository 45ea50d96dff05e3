"""bench.py end to end on the small config-1 workload: one JSON line with the contract's keys."""
import json
import os
import subprocess
import sys

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu


def test_bench_json_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg1", "--steps", "2",
                          "--warmup", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in r, key
    assert r["n_gpus"] == 1 and r["steps"] == 2 and r["warmup"] == 1 and r["higher_is_better"] is True
    # per-step HIP-event times and their median beside the contract's mean over the fenced region
    assert len(r["step_ms"]) == 2 and min(r["step_ms"]) > 0 and min(r["step_ms"]) <= r["median_ms_per_step"] <= max(r["step_ms"])
    assert r["median_ms_per_step"] <= r["ms_per_step"] * 4 and abs(r["value_at_median"] - 3233 * 256 * 256 / (r["median_ms_per_step"] * 1e-3)) < 1e-3 * r["value"]
    assert r["vs_baseline"] is None and r["data"] == "synthetic" and r["value"] > 1e9
    assert "workload" in r["config"] and r["config"]["source_points"] == 3233
    rl = r["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rl, key
    # the bound is the measured limiter (fp32 VALU issue), and a roofline fraction can never exceed 1
    assert rl["bound"] in ("hbm", "mfma") and rl["bound"] == "mfma" and "VALU" in rl["bound_detail"]
    assert rl["unit"] == "TFLOP/s" and rl["peak"] == 157.3
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-9 and 0 < rl["frac"] <= 1
    assert 0 < rl["pipeline"]["frac"] <= 1
    # the memory-side view: top-level keys, labelled for what they are (L2 <-> fabric requests, Infinity-Cache hits included)
    for key in ("fabric_GBs", "fabric_frac", "effective_40B_GBs", "effective_40B_over_peak", "copy_ceiling_GBs", "hbm_peak_GBs"):
        assert key in rl, key
    assert rl["hbm_peak_GBs"] == 8000.0 and "hbm" not in rl
    # the HBM-honest block says whether fabric bytes ARE HBM bytes for this workload (256^2: no, T stays in the Infinity Cache)
    assert rl["hbm_honest"]["hbm_streaming"] is False and "hbm_frac" not in rl["hbm_honest"]
    assert r["target_abs"]["per_gpu"] == 1.2e11 and abs(r["target_abs"]["value_over_target"] - r["value"] / 1.2e11) < 1e-6 * r["value"] / 1.2e11
    assert r["prediction"]["predicted_speedup"] == 1.0
    if rl["traffic"] is not None:
        assert rl["fabric_frac"] <= 1
    # kernel names come from the library (litho_abbe_last_kernels), spelt as rocprofv3 prints them
    assert rl["kernel"].startswith(("k_ypass_", "k_xpass_")) and "<" in rl["kernel"]
    assert rl["kernels"]["xpass"]["kernel"].startswith("k_xpass") and rl["kernels"]["ypass"]["kernel"].startswith("k_ypass")
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["gpu_vs_cpu_rel_to_max"] < 2e-5
    # the parity sample ran the evaluation path of the timed step
    assert cb["parity_path"]["coarse_grid"] == r["config"]["plan"]["coarse_grid"]
    assert "extra_workloads" not in r                           # only the default cfg3 run carries them


@pytest.mark.parametrize("world", [2, 8])
def test_bench_launches_its_own_ranks(world):
    """`python bench.py --gpus N` as a PLAIN process (no torch.distributed.run, no WORLD_SIZE) must start the ranks
    itself (fresh children; the parent never touches the GPU) and relay rank 0's JSON line -- N = 2, and N = 8: the node
    size north_star names, with S = 3233 not divisible by 8.  The test box has one
    GPU and RCCL refuses two ranks on one device, so the ranks share cuda:0 over a gloo group
    (LITHO_BENCH_SHARE_GPU / LITHO_BENCH_BACKEND): everything else -- rendezvous on 127.0.0.1, source-point shards,
    the all-reduce inside abbeImage, barrier + max-over-ranks timing -- is the code the 8-GPU run executes."""
    env = dict(os.environ, LITHO_BENCH_SHARE_GPU="1", LITHO_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--workload", "cfg1", "--steps", "2",
                          "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    shards = [1617, 1616] if world == 2 else [405] + [404] * 7     # contiguous balanced shards (distributed.shard_bounds)
    assert r["n_gpus"] == world and r["scaling"] == "strong" and r["config"]["points_per_rank"] == shards[0]
    assert r["config"]["source_points"] == 3233 and r["value"] > 1e9 and "cpu_baseline" not in r
    assert r["config"]["parallelism"].startswith(f"source-point shards x{world}")
    rk = r["ranks"]                                               # per-rank view of one instrumented step
    assert len(rk["step_ms"]) == world and len(rk["compute_ms"]) == world and len(rk["allreduce_wait_ms"]) == world
    assert rk["source_points"] == shards and sum(shards) == 3233 and rk["allreduce_bytes"] == 256 * 256 * 4
    assert rk["step_ms_max"] >= rk["step_ms_min"] > 0 and min(rk["compute_ms"]) > 0
    # who ran: world size, backend, one identity per rank (here all ranks share cuda:0 -> one distinct device), and the
    # expectation the first real N-GPU record is to be judged against, stated in the record itself
    assert rk["world"] == world and rk["backend"] == "gloo" and rk["rccl_version"] is None
    assert len(rk["devices"]) == world and rk["distinct_devices"] == 1 and "pci" in rk["devices"][0]
    pr = r["prediction"]
    assert pr["predicted_step_ms"] > 0 and 1.0 < pr["predicted_speedup"] <= world and pr["predicted_allreduce_ms"] > 0


def test_bench_adopts_the_launchers_world_size_without_gpus_flag():
    """`torchrun --nproc-per-node 2 bench.py` (no --gpus): the ranks adopt WORLD_SIZE (round-5 advice: --gpus used to default to 1
    and refuse).  Two ranks started by hand the way torch.distributed.run does (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), sharing
    cuda:0 over gloo as above."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, LITHO_BENCH_SHARE_GPU="1", LITHO_BENCH_BACKEND="gloo", RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LITHO_BENCH_TIMEOUT_S="240")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg1", "--steps", "5", "--warmup", "1"],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env))
    outs = [p.communicate(timeout=900) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], [o[1][-1500:] for o in outs]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")]     # rank 0 alone prints
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["ranks"]["world"] == 2 and r["ranks"]["source_points"] == [1617, 1616]
    rk, pr = r["ranks"], r["prediction"]
    assert len(rk["median_step_ms"]) == 2 and r["median_ms_per_step"] == max(rk["median_step_ms"]) and len(r["step_ms"]) == 5
    assert rk["compute_ms_max"] >= rk["compute_ms_min"] > 0 and rk["allreduce_wait_ms_max"] >= rk["allreduce_wait_ms_min"] >= 0
    # the record explains itself: measured over predicted, and where a difference sits
    assert pr["measured_over_predicted"] > 0 and pr["compute_ms_max_over_predicted_compute"] > 0
    assert pr["allreduce_wait_ms_min_over_predicted_allreduce"] is not None


def test_bench_falls_back_to_a_host_staged_group_when_rccl_cannot_start():
    """First contact, last resort: when RCCL cannot be initialised (emulated: LITHO_BENCH_FORCE_RCCL_FAIL=1) the ranks regroup over
    gloo on the next port and the line says so (`ranks.backend_fallback`) -- a labelled record instead of none."""
    env = dict(os.environ, LITHO_BENCH_SHARE_GPU="1", LITHO_BENCH_FORCE_RCCL_FAIL="1")
    for k in ("WORLD_SIZE", "RANK", "LITHO_BENCH_BACKEND"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "cfg1", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 2 and r["ranks"]["backend"] == "gloo" and "FORCE_RCCL_FAIL" in r["ranks"]["backend_fallback"]
    assert r["ranks"]["source_points"] == [1617, 1616] and r["value"] > 1e9
    # and with the fallback switched off the same failure ends the run with the rank named
    env["LITHO_BENCH_NO_FALLBACK"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "cfg1", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode != 0 and "bench.py: rank" in out.stderr


def test_bench_missing_peer_ends_in_seconds_with_the_rank_named():
    """First-contact insurance: WORLD_SIZE = 2 but only rank 0 is ever started (a peer that died before the rendezvous).  The
    rendezvous timeout (LITHO_BENCH_TIMEOUT_S) ends rank 0 with a non-zero code and its name on stderr instead of the lease."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, LITHO_BENCH_SHARE_GPU="1", LITHO_BENCH_BACKEND="gloo", RANK="0", LOCAL_RANK="0", WORLD_SIZE="2",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LITHO_BENCH_TIMEOUT_S="10")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg1", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and "bench.py: rank 0:" in out.stderr and "rendezvous" in out.stderr, out.stderr[-1500:]
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_bench_refuses_a_world_size_other_than_gpus():
    """--gpus N under a launcher that started another number of ranks would report the wrong n_gpus: refused before any GPU work."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "cfg1", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr


def test_bench_rank_failure_does_not_hang():
    """A rank that dies (here: rank 1 asks for a GPU the box does not have) must end the whole launch with a
    non-zero exit code instead of leaving rank 0 waiting at the rendezvous."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a single-GPU box")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "cfg1", "--steps", "1",
                          "--warmup", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode != 0
