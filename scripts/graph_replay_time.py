"""A sequence of images that share pupil and source, replayed from ONE captured HIP graph (torch.cuda.CUDAGraph around
abbeImage with a PlanCache: no host wait inside, so the whole call is capturable): python scripts/graph_replay_time.py [pn] [images]"""
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
cache = L.PlanCache()
for _ in range(5):
    ref = L.abbeImage(mk, mft, pf, bm, 25, mk.deltaK, 193., True, dev, plan_cache=cache)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    L.abbeImage(mk, mft, pf, bm, 25, mk.deltaK, 193., True, dev, plan_cache=cache)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"{pn}^2, S = {S}, PlanCache, eager : {dt * 1e3:.3f} ms per image = {S * pn * pn / dt:.3e} source-pt*px/s")
static_mft = mft.clone()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        L.abbeImage(mk, static_mft, pf, bm, 25, mk.deltaK, 193., True, dev, plan_cache=cache)
torch.cuda.current_stream().wait_stream(side)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    out = L.abbeImage(mk, static_mft, pf, bm, 25, mk.deltaK, 193., True, dev, plan_cache=cache)
graph.replay()
torch.cuda.synchronize()
print("graph replay equals the eager image:", bool(torch.equal(out, ref)))
t0 = time.perf_counter()
for _ in range(n):
    graph.replay()                                  # (a new mask: static_mft.copy_(next spectrum) before the replay)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"{pn}^2, S = {S}, PlanCache, graph : {dt * 1e3:.3f} ms per image = {S * pn * pn / dt:.3e} source-pt*px/s")
