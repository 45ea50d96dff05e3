"""Images per second for a sequence of images that share pupil and source (many masks through one optical setting),
with and without a PlanCache: python scripts/plan_cache_time.py [pn] [images]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lithographysimulator_amd as L
from lithographysimulator_amd.synthetic import bernoulli_mask

pn = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
mk = L.Mask(bernoulli_mask(pn), 25, dev)
mft = mk.fraunhofer(193., True)
bm = L.LightSource(0.0, 0.5, pn, 0.7, device=dev).generateAnnular()
pf = L.Pupil(pn, 193., 0.7, None, dev).generatePupilFunction()
S = int(bm.sum())
for label, cache in (("plain", None), ("PlanCache", L.PlanCache()), ("plain", None), ("PlanCache", L.PlanCache())):
    for _ in range(5):
        L.abbeImage(mk, mft, pf, bm, 25, mk.deltaK, 193., True, dev, plan_cache=cache)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        img = L.abbeImage(mk, mft, pf, bm, 25, mk.deltaK, 193., True, dev, plan_cache=cache)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{pn}^2, S = {S}, {label:9s}: {dt * 1e3:.3f} ms per image = {S * pn * pn / dt:.3e} source-pt*px/s")
