"""BASELINE configs 4 and 5 IN FULL on one GPU against a known answer that involves no transform code: for a mask spectrum of
three isolated orders the Abbe sum is three-beam fringes whose offset and complex contrasts are float64 sums of pupil samples
over the source list (tests/test_gpu_abbe.py::test_few_beam_spectrum_closed_form_full_source has the formula and pins it
against the oracle's op chain).  python scripts/closed_form_full.py [cfg4|cfg5|both]  (about 1 min of GPU each)"""
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lithographysimulator_amd as L                                     # noqa: E402
from lithographysimulator_amd import _native as nat                      # noqa: E402
from test_gpu_abbe import few_beam_closed_form, few_beam_spectrum       # noqa: E402

WL, NA, PS = 193.0, 0.7, 25
DEMO_AB = [0, 0, 0.01, 0, 100, 0.01, 0, 0.01, 0.01, 0.01]
dev = torch.device("cuda", 0)
which = sys.argv[1] if len(sys.argv) > 1 else "both"
amps = [0.9 - 0.2j, 0.4 + 0.8j, -0.6 + 0.1j]


def run(name, pn, pupils, bitmap):
    c = pn // 2
    orders = [(c + pn // 66, c - pn // 27), (c - pn // 15, c + pn // 228), (c + pn // 683, c + pn // 13)]
    eps, N = L.Mask(torch.zeros((pn, pn)), PS, dev).calculateEpsilonN(4 / pn, PS, WL)
    sh = L.sourceShifts(bitmap, pn)
    M = few_beam_spectrum(pn, orders, amps).to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = L.abbeIntensity(M, pupils, sh, N)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    plan = nat.last_plan()
    got = got.cpu().numpy().astype(np.float64)
    stack = pupils if pupils.dim() == 3 else pupils[None]
    worst = 0.0
    for p in range(stack.shape[0]):
        want = few_beam_closed_form(stack[p], sh, orders, amps, pn, N)
        img = got[p] if pupils.dim() == 3 else got
        worst = max(worst, float(np.abs(img - want).max() / want.max()))
    print(f"{name}: {pn}^2 x {stack.shape[0]} plane(s), {sh.shape[0]} source points, {dt:.1f} s on one GPU "
          f"(coarse grid {plan['coarse_grid']}, kernels {nat.last_kernels()}): every pixel against the closed form, "
          f"max error {worst:.2e} of the maximum", flush=True)


if which in ("cfg4", "both"):
    pn = 4096
    run("config 4 in full", pn, L.Pupil(pn, WL, NA, torch.tensor([0, 0, 0, 0, 100], dtype=torch.float16), dev).generatePupilFunction(),
        L.LightSource(0.4, 0.8, pn, NA, device=dev).generateAnnular())
if which in ("cfg5", "both"):
    pn = 2048
    focus = [float(v) for v in np.linspace(-310.0, 310.0, 32)]
    run("config 5 in full", pn, L.throughFocusPupils(pn, WL, NA, torch.tensor(DEMO_AB, dtype=torch.float16), focus, dev),
        L.LightSource(0.4, 0.8, pn, NA, device=dev).generateQuasar(4, -math.pi / 8))
