"""Source-point sharding across ranks on CPU with gloo, world_size 2.  The shard
arithmetic and the single all-reduce are the product's host logic (lithographysimulator_amd/
distributed.py, imageformation.abbeImage); the per-shard intensity is computed here by the CPU
oracle standing in for the HIP kernel, which needs a GPU.  The same path with the real kernels
(two ranks sharing cuda:0) is tests/test_gpu_distributed.py."""
import math
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lithographysimulator_amd.distributed import resolve_group, shard_bounds


@pytest.mark.parametrize("S,world", [(0, 1), (1, 2), (7, 2), (184, 8), (198108, 8), (1581616, 8), (5, 8)])
def test_shard_bounds_partition(S, world):
    spans = [shard_bounds(S, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == S
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 == b0 and a0 <= a1
    sizes = [b - a for a, b in spans]
    assert max(sizes) - min(sizes) <= 1                       # balanced: cost per source point is shift-independent
    if S == 1581616 and world == 8:
        assert sizes == [197702] * 8


def test_resolve_group_is_off_by_default():
    assert resolve_group(None) is None


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LITHO_SHARD_SOURCES="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import abbe_oracle as O
        from lithographysimulator_amd.synthetic import lines_mask
        torch.set_num_threads(2)
        mft = O.mask_spectrum(lines_mask(64), 25, 193.0)
        pf = O.pupil_function(torch.tensor([0, 0, 0.01, 0, 100, 0.01, 0, 0.01, 0.01, 0.01], dtype=torch.float16),
                              64, 0.7, 193.0)
        shifts = O.source_shifts(O.source_quasar(0.4, 0.8, 64, 4, -math.pi / 8), 64)
        eps, N = O.calculate_epsilon_n(4 / 64, 25, 193.0)
        group = resolve_group(None)                            # LITHO_SHARD_SOURCES=1 + initialised default group
        assert group is not None and dist.get_world_size(group) == world
        lo, hi = shard_bounds(shifts.shape[0], dist.get_rank(group), dist.get_world_size(group))
        partial = O.abbe_raw(mft, pf, shifts[lo:hi], N)       # what litho_abbe_accumulate adds on this rank
        dist.all_reduce(partial, op=dist.ReduceOp.SUM, group=group)      # the ONE collective of the path
        image = O.post_process(partial, eps)                  # linear, so it follows the reduction
        if rank == 0:
            whole = O.post_process(O.abbe_raw(mft, pf, shifts, N), eps)
            torch.save({"sharded": image, "whole": whole, "spans": (lo, hi)}, out_path)
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_sum_equals_single_rank(tmp_path):
    out = str(tmp_path / "result.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    assert r["spans"] == (0, 92)
    err = float((r["sharded"] - r["whole"]).abs().max() / r["whole"].max())
    assert err < 2e-6                                          # sharded vs sequential sum = 3.5e-7


def _product_worker(rank, world, port, out_path):
    """The PRODUCT's abbeImage(group=...) with world_size 2: only the three device entry points it calls
    (sourceShifts, abbeIntensity, postProcess) and the device check are replaced by oracle stand-ins."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import lithographysimulator_amd.imageformation as IF
        from lithographysimulator_amd import _native as nat
        from oracle import abbe_oracle as O
        from lithographysimulator_amd.synthetic import lines_mask
        torch.set_num_threads(2)
        calls = []
        nat.require_gpu = lambda d: d if isinstance(d, torch.device) else torch.device(d)
        IF.sourceShifts = lambda bm, pn: O.source_shifts(bm, pn)

        def fake_intensity(mft, pf, shifts, N, out=None, options=None):
            calls.append(int(shifts.shape[0]))
            if pf.dim() == 3:
                return torch.stack([O.abbe_raw(mft, p, shifts, N) if shifts.shape[0] else torch.zeros(mft.shape) for p in pf])
            return O.abbe_raw(mft, pf, shifts, N) if shifts.shape[0] else torch.zeros(mft.shape, dtype=torch.float32)
        IF.abbeIntensity = fake_intensity
        IF.postProcess = lambda raw, eps: (torch.stack([O.post_process(r, eps) for r in raw]) if raw.dim() == 3
                                           else O.post_process(raw, eps))
        mask = IF.Mask(lines_mask(64), 25, torch.device("cpu"))
        mft = O.mask_spectrum(lines_mask(64), 25, 193.0)
        ab = torch.tensor([0, 0, 0.01, 0, 100, 0.01, 0, 0.01, 0.01, 0.01], dtype=torch.float16)
        pf = O.pupil_function(ab.clone(), 64, 0.7, 193.0)
        bm = O.source_quasar(0.4, 0.8, 64, 4, -math.pi / 8)
        cpu = torch.device("cpu")
        res = {}
        res["sharded"] = IF.abbeImage(mask, mft, pf, bm, 25, mask.deltaK, 193.0, True, cpu, group=dist.group.WORLD)
        res["points"] = calls[-1]
        res["normalized"] = IF.abbeImage(mask, mft, pf, bm, 25, mask.deltaK, 193.0, True, cpu, group=dist.group.WORLD,
                                         normalize=True)
        one = torch.zeros_like(bm); one[40, 30] = 1                       # S = 1 < world: rank 1 gets an empty shard
        res["single"] = IF.abbeImage(mask, mft, pf, one, 25, mask.deltaK, 193.0, True, cpu, group=dist.group.WORLD)
        res["single_points"] = calls[-1]
        stack = torch.stack([pf, O.pupil_function(torch.tensor([0, 0, 0, 0, -60], dtype=torch.float16), 64, 0.7, 193.0)])
        res["stack"] = IF.abbeImage(mask, mft, stack, bm, 25, mask.deltaK, 193.0, True, cpu, group=dist.group.WORLD)
        eps, N = O.calculate_epsilon_n(4 / 64, 25, 193.0)
        sh = O.source_shifts(bm, 64)
        res["whole"] = O.post_process(O.abbe_raw(mft, pf, sh, N), eps)
        res["whole_single"] = O.post_process(O.abbe_raw(mft, pf, O.source_shifts(one, 64), N), eps)
        res["whole_stack1"] = O.post_process(O.abbe_raw(mft, stack[1], sh, N), eps)
        res["S"] = int(sh.shape[0])
        torch.save(res, out_path + f".{rank}")
    finally:
        dist.destroy_process_group()


def test_product_abbe_image_with_world_size_two(tmp_path):
    out = str(tmp_path / "result.pt")
    mp.spawn(_product_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    S = r0["S"]
    assert (r0["points"], r1["points"]) == (S - S // 2, S // 2)                # contiguous balanced shards
    assert (r0["single_points"], r1["single_points"]) == (1, 0)                # empty shard on the last rank
    rel = lambda a, b: float((a - b).abs().max() / b.max())
    for r in (r0, r1):                                                         # every rank holds the full image
        assert rel(r["sharded"], r["whole"]) < 2e-6
        assert rel(r["normalized"] * S, r["whole"]) < 2e-6
        assert rel(r["single"], r["whole_single"]) < 2e-6
        assert r["stack"].shape[0] == 2
        assert rel(r["stack"][0], r["whole"]) < 2e-6 and rel(r["stack"][1], r["whole_stack1"]) < 2e-6
    assert torch.equal(r0["sharded"], r1["sharded"])
