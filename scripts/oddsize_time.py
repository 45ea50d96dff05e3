"""Whole-call time per source point for mask sizes that are NOT powers of two, next to the power-of-two size above:
    python scripts/oddsize_time.py [K points]"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lithographysimulator_amd as L
from lithographysimulator_amd import _native as nat
K = int(sys.argv[1]) if len(sys.argv) > 1 else 600
dev = torch.device("cuda", 0)
for pn in (1000, 1024, 1500, 2000, 2048, 768, 3000, 4096):
    gen = torch.Generator().manual_seed(pn)
    geo = (torch.rand(pn, pn, generator=gen) < 0.5).to(torch.int16)
    mask = L.Mask(geo, 25, dev); mft = mask.fraunhofer(193., True)
    eps, N = mask.calculateEpsilonN(mask.deltaK, 25, 193.)
    pf = L.Pupil(pn, 193., 0.7, torch.tensor([0, 0, 0.01, 0, 100, 0.01, 0, 0.01, 0.01, 0.01], dtype=torch.float16), dev).generatePupilFunction()
    sh = L.sourceShifts(L.LightSource(0.4, 0.8, pn, 0.7, device=dev).generateAnnular(), pn)
    k = min(K, sh.shape[0] // 2) if pn < 3000 else min(K // 4, sh.shape[0] // 2)
    sel = sh[sh.shape[0] // 3: sh.shape[0] // 3 + k].contiguous()
    L.abbeIntensity(mft, pf, sel, N)
    best = 1e30
    for _ in range(3):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); L.abbeIntensity(mft, pf, sel, N); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    p = nat.last_plan()
    print(f"pn {pn:5d} N {N:5d}  {best * 1e3 / k:8.3f} us/pt  = {k * pn * pn / best / 1e-3:.3e} pt*px/s   variant {p['variant']} coarse {p['coarse_grid']} wave {p['wave_ypass']} box {p['box_rows']}x{p['box_cols']} kernels {nat.last_kernels()}", flush=True)
