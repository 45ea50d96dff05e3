"""Embedded (padded to N/2 or N) against un-embedded evaluation, whole-call us per source point:
    python scripts/embed_ab.py pn:pixelSize [pn:pixelSize ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lithographysimulator_amd as L
from lithographysimulator_amd import _native as nat
dev = torch.device("cuda", 0)
for arg in sys.argv[1:]:
    pn, ps = (int(v) for v in arg.split(":"))
    gen = torch.Generator().manual_seed(pn)
    mask = L.Mask((torch.rand(pn, pn, generator=gen) < 0.5).to(torch.int16), ps, dev); mft = mask.fraunhofer(193., True)
    eps, N = mask.calculateEpsilonN(mask.deltaK, ps, 193.)
    pf = L.Pupil(pn, 193., 0.7, torch.tensor([0, 0, 0.01, 0, 100, 0.01], dtype=torch.float16), dev).generatePupilFunction()
    sh = L.sourceShifts(L.LightSource(0.4, 0.8, pn, 0.7, device=dev).generateAnnular(), pn)
    k = min(600 if pn <= 2048 else 150, sh.shape[0] // 2)
    sel = sh[sh.shape[0] // 3: sh.shape[0] // 3 + k].contiguous()
    res = {}
    for emb in (True, False, True, False):
        o = {"embed": int(emb)}
        L.abbeIntensity(mft, pf, sel, N, options=o)
        best = 1e30
        for _ in range(3):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); L.abbeIntensity(mft, pf, sel, N, options=o); b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        p = nat.last_plan()
        res[emb] = min(res.get(emb, 1e30), best * 1e3 / k)
        last = (p["variant"], p["coarse_grid"], nat.last_kernels())
        if emb: lastE = last
        else: lastP = last
    print(f"pn {pn} ps {ps} N {N} -> {L.embeddedSize(pn, N)}: embedded {res[True]:8.3f} us/pt {lastE} | plain {res[False]:8.3f} us/pt {lastP}", flush=True)
