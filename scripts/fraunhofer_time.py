"""Mask.fraunhofer (litho_mask_spectrum) time per mask next to the image time of the same size."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lithographysimulator_amd as L
from lithographysimulator_amd.synthetic import bernoulli_mask
dev = torch.device("cuda", 0)
for pn in (256, 512, 1000, 1024, 2048, 4096):
    mask = L.Mask(bernoulli_mask(pn), 25, dev)
    mask.fraunhofer(193., True); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): mask.fraunhofer(193., True)
    b.record(); torch.cuda.synchronize()
    print(f"pn {pn:5d}: fraunhofer {a.elapsed_time(b) / 20 * 1e3:9.1f} us per mask", flush=True)
