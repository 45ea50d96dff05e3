"""Break-even source count of the coarse-grid path (its reconstruction is a fixed cost per call and plane):
    python scripts/coarse_breakeven.py [pn ...]
Whole-call time (HIP events, best of 5) of abbeIntensity over S consecutive source points, direct (coarse=0) against
coarse grid (coarse=2), S doubling; prints both and the ratio.  The planner's thresholds (abbe_engine.hip: s_min) come from here."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lithographysimulator_amd as L                                     # noqa: E402
from lithographysimulator_amd.synthetic import bernoulli_mask            # noqa: E402

dev = torch.device("cuda", 0)
for pn in [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2048, 4096]:
    mask = L.Mask(bernoulli_mask(pn), 25, dev)
    mft = mask.fraunhofer(193., True)
    eps, N = mask.calculateEpsilonN(mask.deltaK, 25, 193.)
    pf = L.Pupil(pn, 193., 0.7, torch.tensor([0, 0, 0.01, 0, 100, 0.01, 0, 0.01, 0.01, 0.01], dtype=torch.float16), dev).generatePupilFunction()
    sh = L.sourceShifts(L.LightSource(0.0, 0.9, pn, 0.7, device=dev).generateAnnular(), pn)
    lo = sh.shape[0] // 3
    S = 32
    while S <= min(sh.shape[0] - lo, {256: 32768, 512: 8192, 1024: 2048, 2048: 512, 4096: 256}.get(pn, 512)):
        sel = sh[lo:lo + S].contiguous()
        t = {}
        for c in (0, 2):
            L.abbeIntensity(mft, pf, sel, N, options={"coarse": c})
            best = 1e30
            for _ in range(5):
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); L.abbeIntensity(mft, pf, sel, N, options={"coarse": c}); b.record(); torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b))
            t[c] = best
        print(f"pn {pn:5d}  S {S:6d}   direct {t[0] * 1e3:9.1f} us   coarse {t[2] * 1e3:9.1f} us   coarse/direct {t[2] / t[0]:.3f}", flush=True)
        S *= 2
