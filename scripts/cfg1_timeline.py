"""Config 1 (256^2, S = 3233) image after image, for a rocprofv3 --kernel-trace timeline (round-5 review item 7: where do the
0.55 ms of an unplanned image go?):
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cfg1_trace -o t -- python3 scripts/cfg1_timeline.py 40 plain
    python3 scripts/trace_summary.py gpurun_out/cfg1_trace/*/t_kernel_trace.csv 20
mode: plain | planned (one PlanCache for the whole sequence)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lithographysimulator_amd as L
from lithographysimulator_amd.synthetic import bernoulli_mask

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
mode = sys.argv[2] if len(sys.argv) > 2 else "plain"
pn = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda", 0)
mk = L.Mask(bernoulli_mask(pn), 25, dev)
mft = mk.fraunhofer(193., True)
bm = L.LightSource(0.0, 0.5, pn, 0.7, device=dev).generateAnnular()
pf = L.Pupil(pn, 193., 0.7, None, dev).generatePupilFunction()
S = int(bm.sum())
cache = L.PlanCache() if mode == "planned" else None
for _ in range(5):
    L.abbeImage(mk, mft, pf, bm, 25, mk.deltaK, 193., True, dev, plan_cache=cache)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    img = L.abbeImage(mk, mft, pf, bm, 25, mk.deltaK, 193., True, dev, plan_cache=cache)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"{pn}^2, S = {S}, {mode}: {dt * 1e3:.3f} ms per image = {S * pn * pn / dt:.3e} source-pt*px/s", file=sys.stderr)
