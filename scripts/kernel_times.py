"""Per-kernel HIP-event times (x-pass / y-pass) for the libs given: python scripts/kernel_times.py pn K lib1,lib2"""
import math, os, subprocess, sys
import os as _os; _os.environ.setdefault("LITHO_ABBE_COARSE", "2")   # timing probes use short source lists: do not let the S threshold pick the direct path silently
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, ROOT)
    import lithographysimulator_amd as L
    from lithographysimulator_amd import _native as nat
    from lithographysimulator_amd.synthetic import bernoulli_mask
    pn, K = int(sys.argv[2]), int(sys.argv[3])
    dev = torch.device("cuda", 0)
    mask = L.Mask(bernoulli_mask(pn), 25, dev); mft = mask.fraunhofer(193., True)
    eps, N = mask.calculateEpsilonN(mask.deltaK, 25, 193.)
    pf = L.Pupil(pn, 193., 0.7, torch.tensor([0, 0, 0.01, 0, 100, 0.01, 0, 0.01, 0.01, 0.01], dtype=torch.float16), dev).generatePupilFunction()
    sh = L.sourceShifts(L.LightSource(0.4, 0.8, pn, 0.7, device=dev).generateQuasar(4, -math.pi / 8), pn)
    sel = sh[(torch.arange(K, device=dev) * sh.shape[0]) // K].contiguous()
    L.abbeIntensity(mft, pf, sel, N)
    nat.set_profiling(True)
    best = None
    for _ in range(3):
        L.abbeIntensity(mft, pf, sel, N); torch.cuda.synchronize()
        p = nat.last_profile()
        cur = (p["xpass_ms"] / p["xpass_points"] * 1e3, p["ypass_ms"] / p["ypass_points"] * 1e3)
        best = cur if best is None or sum(cur) < sum(best) else best
    print(f"   x-pass {best[0]:6.2f} us/pt   y-pass {best[1]:6.2f} us/pt   sum {sum(best):6.2f}   plan batch {nat.last_plan()['batch']}", flush=True)
    sys.exit(0)
pn, K = sys.argv[1], sys.argv[2]
for lib in sys.argv[3].split(","):
    env = dict(os.environ, LITHO_ALLOW_DIAG="1")
    if lib != "default": env["LITHO_ABBE_LIB"] = os.path.join(ROOT, "build", "variants", f"lib_{lib}.so")
    print(f"== {lib}", flush=True)
    subprocess.run([sys.executable, __file__, "--child", pn, K], env=env)
