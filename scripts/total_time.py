"""Wall-clock (HIP events around the call) of one abbeIntensity over K consecutive source points:
    python scripts/total_time.py pn K [planes]        -> us per source point (and plane), best of 3"""
import math, os, sys, torch
import os as _os; _os.environ.setdefault("LITHO_ABBE_COARSE", "2")   # timing probes use short source lists: do not let the S threshold pick the direct path silently
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lithographysimulator_amd as L
from lithographysimulator_amd import _native as nat
from lithographysimulator_amd.synthetic import bernoulli_mask
pn, K = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
mask = L.Mask(bernoulli_mask(pn), 25, dev); mft = mask.fraunhofer(193., True)
eps, N = mask.calculateEpsilonN(mask.deltaK, 25, 193.)
ab = torch.tensor([0, 0, 0.01, 0, 100, 0.01, 0, 0.01, 0.01, 0.01], dtype=torch.float16)
pf = L.Pupil(pn, 193., 0.7, ab, dev).generatePupilFunction()
sh = L.sourceShifts(L.LightSource(0.4, 0.8, pn, 0.7, device=dev).generateQuasar(4, -math.pi / 8), pn)
sel = sh[sh.shape[0] // 3: sh.shape[0] // 3 + K].contiguous()
ref = None
if os.environ.get("CHECK"):
    env = {k: os.environ.pop(k) for k in list(os.environ) if k.startswith("LITHO_ABBE_OVERLAP")}
    ref = L.abbeIntensity(mft, pf, sel, N).clone()
    os.environ.update(env)
out = L.abbeIntensity(mft, pf, sel, N)
best = 1e30
for _ in range(3):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); out = L.abbeIntensity(mft, pf, sel, N); b.record(); torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b))
msg = f"pn={pn} K={K}: {best * 1e3 / K:7.3f} us/pt total  batch {nat.last_plan()['batch']}"
if ref is not None:
    msg += f"  vs serial rel-to-max {float((out - ref).abs().max() / ref.max()):.2e}"
print(msg)
