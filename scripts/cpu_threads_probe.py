"""Which torch thread count gives the best CPU-port throughput on this host (for a fair cpu_baseline)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import abbe_oracle as O
pn, N = 2048, 4096
g = torch.Generator().manual_seed(0)
m = torch.complex(torch.randn(pn, pn, generator=g), torch.randn(pn, pn, generator=g))
p = torch.complex(torch.randn(pn, pn, generator=g), torch.randn(pn, pn, generator=g))
sh = torch.tensor([[3, -5], [10, 7]], dtype=torch.int32)
for nt in (8, 16, 32, 64, 128, 256):
    if nt > (os.cpu_count() or 1): break
    torch.set_num_threads(nt)
    O.abbe_raw(m, p, sh[:1], N)
    t0 = time.perf_counter(); O.abbe_raw(m, p, sh, N); dt = (time.perf_counter() - t0) / 2
    print(f"threads={nt} {dt:.3f} s/pt {pn*pn/dt:.3e} pt*px/s", flush=True)
