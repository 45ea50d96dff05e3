"""The RCCL leg of the path on the GPU box: abbeImage(..., group=...) with backend "nccl" (= RCCL on ROCm).
Only one GPU is available to the tests, so the group has one rank; the code path (shard bounds, accumulate on
the shard, all_reduce over RCCL, post-process) is the one bench.py --gpus N uses.  World-size-2 behaviour of
the sharding arithmetic is covered on CPU with gloo (tests/test_distributed_cpu.py)."""
import os
import subprocess
import sys
import textwrap

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu

SCRIPT = textwrap.dedent("""
    import math, os, sys, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
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
    out = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][0].split()
    assert float(line[1]) == 0.0 and float(line[2]) == 0.0
    assert abs(float(line[3]) / 2.2029254e13 - 1) < 1e-5            # the reference's demo image sum
