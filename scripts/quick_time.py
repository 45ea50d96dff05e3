"""Quick device-time probe of litho_abbe_accumulate (not the bench; for tuning)."""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lithographysimulator_amd as L
from lithographysimulator_amd import _native as nat
from lithographysimulator_amd.synthetic import bernoulli_mask

pn = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda", 0)
mask = L.Mask(bernoulli_mask(pn), 25, dev)
mft = mask.fraunhofer(193., True)
eps, N = mask.calculateEpsilonN(mask.deltaK, 25, 193.)
ab = torch.tensor([0, 0, 0.01, 0, 100, 0.01, 0, 0.01, 0.01, 0.01], dtype=torch.float16)
pf = L.Pupil(pn, 193., 0.7, ab, dev).generatePupilFunction()
bm = L.LightSource(0.4, 0.8, pn, 0.7, device=dev).generateQuasar(4, -math.pi / 8)
sh = L.sourceShifts(bm, pn)
S = sh.shape[0]
sel = sh[(torch.arange(K, device=dev) * S) // K].contiguous()
for trial in range(3):
    torch.cuda.synchronize()
    t0 = time.time()
    out = L.abbeIntensity(mft, pf, sel, N)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"pn={pn} N={N} S={S} K={K} t={dt*1e3:.1f} ms  {dt/K*1e6:.1f} us/pt  {K*pn*pn/dt:.3e} pt*px/s  "
          f"frac40B={40*K*pn*pn/dt/8e12:.3f}  plan={nat.last_plan()}", flush=True)
