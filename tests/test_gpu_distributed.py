"""The RCCL leg of the path on the GPU box: abbeImage(..., group=...) with backend "nccl" (= RCCL on ROCm).
On a one-GPU box the RCCL group has one rank; the code path (shard bounds, accumulate on the shard, all_reduce over
RCCL, post-process) is the one bench.py --gpus N uses.  The world-size-2 behaviour of that same path with the real
kernels is exercised by two ranks SHARING cuda:0 over gloo (second test), and on CPU with oracle stand-ins
(tests/test_distributed_cpu.py).  On a box with SEVERAL GPUs the last tests run the real thing -- world size 2 .. N,
one rank per device, RCCL all-reduce -- and skip themselves elsewhere."""
import os
import subprocess
import sys
import textwrap

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu

def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


SCRIPT = textwrap.dedent("""
    import math, os, sys, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    os.environ.update(MASTER_ADDR="127.0.0.1", RANK="0", WORLD_SIZE="1")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    import lithographysimulator_amd as L
    m = L.Mask(device=dev); mft = m.fraunhofer(193., True)
    ls = L.LightSource(0.4, 0.8, device=dev).generateQuasar(4, -math.pi / 8)
    pf = L.Pupil(64, 193., 0.7, torch.tensor([0, 0, 0.01, 0, 100, 0.01, 0, 0.01, 0.01, 0.01], dtype=torch.float16), dev).generatePupilFunction()
    a = L.abbeImage(m, mft, pf, ls, 25, m.deltaK, 193., True, dev, group=dist.group.WORLD)
    os.environ["LITHO_SHARD_SOURCES"] = "1"
    b = L.abbeImage(m, mft, pf, ls, 25, m.deltaK, 193., True, dev)          # default group picked up from the env
    del os.environ["LITHO_SHARD_SOURCES"]
    c = L.abbeImage(m, mft, pf, ls, 25, m.deltaK, 193., True, dev)          # no group at all
    torch.cuda.synchronize()
    print("RESULT", float((a - c).abs().max() / c.max()), float((b - c).abs().max() / c.max()), float(c.sum()))
    dist.destroy_process_group()
""") % ROOT


def test_rccl_group_path_single_rank():
    out = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=600, cwd=ROOT,
                         env=dict(os.environ, MASTER_PORT=str(_free_port())))
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][0].split()
    assert float(line[1]) == 0.0 and float(line[2]) == 0.0
    assert abs(float(line[3]) / 2.2029254e13 - 1) < 1e-5            # the reference's demo image sum


# Two FRESH processes (started before anything here touches the GPU), both on cuda:0, gloo rendezvous: the product's
# abbeImage(group=WORLD) = shard -> HIP accumulate on the shard -> ONE all-reduce -> post-process, world_size 2.
# (RCCL refuses two ranks on one device, so the collective of this test is gloo's; the 8-GPU run uses "nccl".)
TWO_RANK = textwrap.dedent("""
    import math, os, sys, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    rank = int(os.environ["RANK"])
    dist.init_process_group("gloo", rank=rank, world_size=2)
    dev = torch.device("cuda", 0)
    import lithographysimulator_amd as L
    from lithographysimulator_amd import _native as nat
    from lithographysimulator_amd.synthetic import bernoulli_mask
    W = dist.group.WORLD
    res = {}
    # (a) BASELINE config 1 end to end: 256^2, S = 3233 (odd: shards of 1617 and 1616)
    m = L.Mask(bernoulli_mask(256), 25, dev); mft = m.fraunhofer(193., True)
    bm = L.LightSource(0.0, 0.5, 256, 0.7, device=dev).generateAnnular()
    pf = L.Pupil(256, 193., 0.7, None, dev).generatePupilFunction()
    sharded = L.abbeImage(m, mft, pf, bm, 25, m.deltaK, 193., True, dev, group=W)
    whole = L.abbeImage(m, mft, pf, bm, 25, m.deltaK, 193., True, dev)
    res["a"] = float((sharded - whole).abs().max() / whole.max())
    # (b) S = 1 < world: the last rank's shard is empty
    one = torch.zeros_like(bm); one[140, 120] = 1
    s1 = L.abbeImage(m, mft, pf, one, 25, m.deltaK, 193., True, dev, group=W)
    w1 = L.abbeImage(m, mft, pf, one, 25, m.deltaK, 193., True, dev)
    res["b"] = float((s1 - w1).abs().max() / w1.max())
    # (c) a 3-plane through-focus stack at 1024^2 (fused x-pass), 22 strided annular points, rank-dependent shard sizes
    m2 = L.Mask(bernoulli_mask(1024), 25, dev); mft2 = m2.fraunhofer(193., True)
    ab = torch.tensor([0, 0, 0.01, 0, 100, 0.01, 0, 0.01, 0.01, 0.01], dtype=torch.float16)
    stack = L.throughFocusPupils(1024, 193., 0.7, ab, [-110.0, 10.0, 90.0], dev)
    full = L.LightSource(0.4, 0.8, 1024, 0.7, device=dev).generateAnnular()
    pts = torch.argwhere(full); idx = (torch.arange(21, device=dev) * pts.shape[0]) // 21
    sub = torch.zeros_like(full); sub[pts[idx, 0], pts[idx, 1]] = 1
    s3 = L.abbeImage(m2, mft2, stack, sub, 25, m2.deltaK, 193., True, dev, group=W)
    w3 = L.abbeImage(m2, mft2, stack, sub, 25, m2.deltaK, 193., True, dev)
    res["c"] = float((s3 - w3).abs().max() / w3.max()); res["c_shape"] = list(s3.shape)
    torch.cuda.synchronize()
    print("RESULT", rank, res["a"], res["b"], res["c"], res["c_shape"], float(whole.double().sum()))
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def test_two_ranks_share_one_gpu_over_gloo(golden):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, "-c", TWO_RANK], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              cwd=ROOT, env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                                                 RANK=str(r), WORLD_SIZE="2"))
             for r in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    ref_sum = float(golden("g5_images.npz")["cfg1_bern_final"].astype("float64").sum())
    for so, _ in outs:
        f = [l for l in so.splitlines() if l.startswith("RESULT")][0].split(maxsplit=5)
        a, b, c = float(f[2]), float(f[3]), float(f[4])
        assert a < 2e-6 and b < 2e-6 and c < 2e-6, (a, b, c)
        assert "[3, 1024, 1024]" in f[5]
        assert abs(float(f[5].split("]")[1]) / ref_sum - 1) < 1e-5          # and it is the reference's config-1 image


# ---------------------------------------------------------------------------------------------------------------
# RCCL with more than one rank, one rank per DEVICE (needs >= 2 GPUs: skipped on the one-GPU test box, runs on the
# 8-GPU node).  Fresh child processes started before anything here touches a GPU; free port on 127.0.0.1.
# ---------------------------------------------------------------------------------------------------------------
MULTI_RANK = textwrap.dedent("""
    import math, os, sys, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)                # RCCL over xGMI
    import lithographysimulator_amd as L
    from lithographysimulator_amd.synthetic import bernoulli_mask
    W = dist.group.WORLD
    # (a) BASELINE config 1 end to end (S = 3233: uneven shards for every world size that does not divide it)
    m = L.Mask(bernoulli_mask(256), 25, dev); mft = m.fraunhofer(193., True)
    bm = L.LightSource(0.0, 0.5, 256, 0.7, device=dev).generateAnnular()
    pf = L.Pupil(256, 193., 0.7, None, dev).generatePupilFunction()
    sharded = L.abbeImage(m, mft, pf, bm, 25, m.deltaK, 193., True, dev, group=W)
    whole = L.abbeImage(m, mft, pf, bm, 25, m.deltaK, 193., True, dev)
    a = float((sharded - whole).abs().max() / whole.max())
    # every rank must hold the SAME reduced image: compare against rank 0's copy
    ref0 = sharded.clone(); dist.broadcast(ref0, 0)
    same = float((sharded - ref0).abs().max())
    # (b) fewer source points than ranks: some shards are empty
    few = torch.zeros_like(bm); few[140, 120] = 1
    if world > 2: few[100, 131] = 1
    s1 = L.abbeImage(m, mft, pf, few, 25, m.deltaK, 193., True, dev, group=W)
    w1 = L.abbeImage(m, mft, pf, few, 25, m.deltaK, 193., True, dev)
    b = float((s1 - w1).abs().max() / w1.max())
    # (c) a 3-plane through-focus stack at 1024^2, normalised: the 48 MiB all-reduce of a stack
    m2 = L.Mask(bernoulli_mask(1024), 25, dev); mft2 = m2.fraunhofer(193., True)
    ab = torch.tensor([0, 0, 0.01, 0, 100, 0.01, 0, 0.01, 0.01, 0.01], dtype=torch.float16)
    stack = L.throughFocusPupils(1024, 193., 0.7, ab, [-110.0, 10.0, 90.0], dev)
    full = L.LightSource(0.4, 0.8, 1024, 0.7, device=dev).generateAnnular()
    pts = torch.argwhere(full); idx = (torch.arange(41, device=dev) * pts.shape[0]) // 41
    sub = torch.zeros_like(full); sub[pts[idx, 0], pts[idx, 1]] = 1
    s3 = L.abbeImage(m2, mft2, stack, sub, 25, m2.deltaK, 193., True, dev, group=W, normalize=True)
    w3 = L.abbeImage(m2, mft2, stack, sub, 25, m2.deltaK, 193., True, dev, normalize=True)
    c = float((s3 - w3).abs().max() / w3.max())
    torch.cuda.synchronize()
    print("RESULT", rank, a, b, c, same, float(whole.double().sum()))
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def _gpu_count():
    import torch
    return torch.cuda.device_count()          # does not initialise the GPU on this image


def _run_ranks(script, world, timeout=900):
    port = _free_port()
    env0 = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world),
                HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", script], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              cwd=ROOT, env=dict(env0, RANK=str(r), LOCAL_RANK=str(r))) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return procs, outs


@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_multi_rank_one_device_each(golden, world):
    if _gpu_count() < world:
        pytest.skip(f"needs {world} GPUs (this box has {_gpu_count()})")
    procs, outs = _run_ranks(MULTI_RANK, world)
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    ref_sum = float(golden("g5_images.npz")["cfg1_bern_final"].astype("float64").sum())
    for so, _ in outs:
        f = [l for l in so.splitlines() if l.startswith("RESULT")][0].split()
        a, b, c, same, total = (float(v) for v in f[2:7])
        assert a < 2e-6 and b < 2e-6 and c < 2e-6, (a, b, c)
        assert same == 0.0                                           # all ranks hold the identical all-reduced image
        assert abs(total / ref_sum - 1) < 1e-5


def test_bench_multi_gpu_rccl_line():
    """bench.py --gpus N over RCCL on distinct devices (the command the driver's scaling run issues), smallest workload:
    the JSON line must carry the per-rank view."""
    import json
    n = min(_gpu_count(), 8)
    if n < 2:
        pytest.skip("needs at least 2 GPUs")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LITHO_BENCH_SHARE_GPU", "LITHO_BENCH_BACKEND"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--workload", "cfg2", "--steps", "2",
                          "--warmup", "1"], capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert r["n_gpus"] == n and r["config"]["source_points"] == 98832 and len(r["ranks"]["compute_ms"]) == n
    assert sum(r["ranks"]["source_points"]) == 98832 and r["ranks"]["allreduce_bytes"] == 1024 * 1024 * 4
