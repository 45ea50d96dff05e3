"""Device-time probe of a through-focus stack (config-5 geometry): python scripts/stack_time.py pn planes K
Prints us per T item (source point x plane) for the x-pass and the y-pass; knobs via LITHO_ABBE_* env."""
import math
import os as _os; _os.environ.setdefault("LITHO_ABBE_COARSE", "2")   # timing probes use short source lists: do not let the S threshold pick the direct path silently
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lithographysimulator_amd as L
from lithographysimulator_amd import _native as nat
from lithographysimulator_amd.synthetic import bernoulli_mask

pn = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
planes = int(sys.argv[2]) if len(sys.argv) > 2 else 8
K = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dev = torch.device("cuda", 0)
mask = L.Mask(bernoulli_mask(pn), 25, dev)
mft = mask.fraunhofer(193., True)
eps, N = mask.calculateEpsilonN(mask.deltaK, 25, 193.)
ab = torch.tensor([0, 0, 0.01, 0, 100, 0.01, 0, 0.01, 0.01, 0.01], dtype=torch.float16)
defocus = [-310.0 + 20 * k for k in range(planes)]
stack = L.throughFocusPupils(pn, 193., 0.7, ab, defocus, dev) if planes > 1 else L.Pupil(pn, 193., 0.7, ab, dev).generatePupilFunction()
sh = L.sourceShifts(L.LightSource(0.4, 0.8, pn, 0.7, device=dev).generateQuasar(4, -math.pi / 8), pn)
sel = sh[sh.shape[0] // 3: sh.shape[0] // 3 + K].contiguous()        # CONSECUTIVE points, as in the bench
L.abbeIntensity(mft, stack, sel, N)
nat.set_profiling(True)
best = None
for _ in range(3):
    L.abbeIntensity(mft, stack, sel, N)
    torch.cuda.synchronize()
    p = nat.last_profile()
    cur = (p["xpass_ms"] / p["xpass_points"] * 1e3, p["ypass_ms"] / p["ypass_points"] * 1e3)
    best = cur if best is None or sum(cur) < sum(best) else best
print(f"pn={pn} planes={planes} K={K}: x-pass {best[0]:6.2f} us/item  y-pass {best[1]:6.2f} us/item  sum {sum(best):6.2f}  "
      f"plan={nat.last_plan()}", flush=True)
