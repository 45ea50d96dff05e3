"""General (wrapping) mode against the pruned path: whole-call us per source point for a centred and a shifted annular source
(the shifted one wraps the pupil around the grid for part of its points): python scripts/wrap_time.py [pn ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lithographysimulator_amd as L
from lithographysimulator_amd import _native as nat
from lithographysimulator_amd.synthetic import bernoulli_mask
dev = torch.device("cuda", 0)
for pn in [int(a) for a in sys.argv[1:]] or [1024, 2048]:
    mask = L.Mask(bernoulli_mask(pn), 25, dev); mft = mask.fraunhofer(193., True)
    eps, N = mask.calculateEpsilonN(mask.deltaK, 25, 193.)
    pf = L.Pupil(pn, 193., 0.7, torch.tensor([0, 0, 0.01, 0, 100, 0.01], dtype=torch.float16), dev).generatePupilFunction()
    for name, sx, sy in (("centred", 0.0, 0.0), ("shifted 0.25/-0.5", 0.25, -0.5)):
        sh = L.sourceShifts(L.LightSource(0.4, 0.8, pn, 0.7, shiftX=sx, shiftY=sy, device=dev).generateAnnular(), pn)
        k = min(1200 if pn <= 1024 else 300, sh.shape[0])
        sel = sh[(torch.arange(k, device=dev) * sh.shape[0]) // k].contiguous()
        c, h = pn // 2, pn // 4
        wraps = int(((sel[:, 0] < -(c - h)) | (sel[:, 0] > (pn - 1) - (c + h)) | (sel[:, 1] < -(c - h)) | (sel[:, 1] > (pn - 1) - (c + h))).sum())
        L.abbeIntensity(mft, pf, sel, N)
        best = 1e30
        for _ in range(3):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); L.abbeIntensity(mft, pf, sel, N); b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        p = nat.last_plan()
        print(f"pn {pn} {name:18s}: {best * 1e3 / k:8.3f} us/pt  general {p['general']} coarse {p['coarse_grid']}  points that wrap: {wraps} of {k}  kernels {nat.last_kernels()}", flush=True)
