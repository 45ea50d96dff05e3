"""A pupil wider than the natural box |k| <= pn/4 (custom / apodised pupils): the generic kernels at the caller's size against
the same problem padded by hand into the N-point grid (where every pupil on the pn grid lies inside the natural box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lithographysimulator_amd as L
from lithographysimulator_amd import _native as nat
from lithographysimulator_amd.synthetic import bernoulli_mask
dev = torch.device("cuda", 0)
for pn in (1024, 2048):
    mask = L.Mask(bernoulli_mask(pn), 25, dev); mft = mask.fraunhofer(193., True)
    eps, N = mask.calculateEpsilonN(mask.deltaK, 25, 193.)
    yy, xx = torch.meshgrid(torch.arange(pn) - pn // 2, torch.arange(pn) - pn // 2, indexing="ij")
    inside = (yy.abs() < 0.35 * pn) & (xx.abs() < 0.35 * pn)
    pupil = torch.where(inside, torch.polar(torch.exp(-(xx ** 2 + yy ** 2) / (0.3 * pn) ** 2), 0.002 * (xx * yy).float() / pn), torch.zeros((), dtype=torch.complex64)).to(torch.complex64).to(dev)
    sh = L.sourceShifts(L.LightSource(0.0, 0.5, pn, 0.7, device=dev).generateAnnular(), pn)
    k = 600 if pn == 1024 else 150
    sel = sh[(torch.arange(k, device=dev) * sh.shape[0]) // k].contiguous()
    def timeit(fn):
        fn(); best = 1e30
        for _ in range(3):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); r = fn(); b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        return best * 1e3 / k, r
    t0, r0 = timeit(lambda: L.abbeIntensity(mft, pupil, sel, N))
    p0 = (nat.last_plan()["variant"], nat.last_plan()["general"], nat.last_kernels())
    pe = N; o = (pe - pn) // 2
    m2 = torch.zeros(pe, pe, dtype=torch.complex64, device=dev); m2[o:o + pn, o:o + pn] = mft
    p2 = torch.zeros(pe, pe, dtype=torch.complex64, device=dev); p2[o:o + pn, o:o + pn] = pupil
    t1, r1 = timeit(lambda: L.abbeIntensity(m2, p2, sel, N))
    p1 = (nat.last_plan()["variant"], nat.last_plan()["general"], nat.last_kernels())
    d = float((r1[o:o + pn, o:o + pn] - r0).abs().max() / r0.max())
    print(f"pn {pn}: generic {t0:8.3f} us/pt {p0} | padded into {pe} {t1:8.3f} us/pt {p1} | diff {d:.1e}", flush=True)
