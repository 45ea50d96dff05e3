"""Alternating A/B of launch-planner OPTIONS on one box, per-kernel HIP-event times over consecutive source points:
    python scripts/option_ab.py <pn> <K points> <reps> "coarse=2" "coarse=2,xstorewave=1" ...
Prints x-pass / y-pass microseconds per source point, the kernels that ran, and max|diff| / max of the image against the
first option set (0 = bit-identical)."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lithographysimulator_amd as L                                     # noqa: E402
from lithographysimulator_amd import _native as nat                      # noqa: E402
from lithographysimulator_amd.synthetic import bernoulli_mask            # noqa: E402

pn, K, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sets = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",") if kv) for a in sys.argv[4:]]
dev = torch.device("cuda", 0)
mask = L.Mask(bernoulli_mask(pn), 25, dev)
mft = mask.fraunhofer(193., True)
eps, N = mask.calculateEpsilonN(mask.deltaK, 25, 193.)
pf = L.Pupil(pn, 193., 0.7, torch.tensor([0, 0, 0.01, 0, 100, 0.01, 0, 0.01, 0.01, 0.01], dtype=torch.float16), dev).generatePupilFunction()
ls = L.LightSource(0.4, 0.8, pn, 0.7, device=dev)
sh = L.sourceShifts(ls.generateQuasar(4, -math.pi / 8) if pn == 2048 else ls.generateAnnular(), pn)
lo = sh.shape[0] // 3
sel = sh[lo:lo + K].contiguous()
ref = None
for o in sets:
    L.abbeIntensity(mft, pf, sel[:64], N, options=o)                     # warm
for r in range(reps):
    for o in sets:
        nat.set_profiling(True)
        img = L.abbeIntensity(mft, pf, sel, N, options=o)
        torch.cuda.synchronize()
        p = nat.last_profile()
        nat.set_profiling(False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); L.abbeIntensity(mft, pf, sel, N, options=o); e1.record(); torch.cuda.synchronize()
        if ref is None:
            ref = img.clone()
        d = float((img - ref).abs().max() / ref.max())
        print(f"{str(o):44s} x {p['xpass_ms'] / p['xpass_points'] * 1e3:6.3f}  y {p['ypass_ms'] / p['ypass_points'] * 1e3:6.3f} us/pt | "
              f"whole call {e0.elapsed_time(e1) / K * 1e3:6.3f} us/pt | {p['xpass_kernel']} / {p['ypass_kernel']} | diff {d:.1e}", flush=True)
