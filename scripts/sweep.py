"""Sweep runtime knobs (and, via LITHO_ABBE_LIB, compile-time variants) of the Abbe engine."""
import itertools, math, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, ROOT)
    import lithographysimulator_amd as L
    from lithographysimulator_amd import _native as nat
    from lithographysimulator_amd.synthetic import bernoulli_mask
    pn, K = int(sys.argv[2]), int(sys.argv[3])
    dev = torch.device("cuda", 0)
    mask = L.Mask(bernoulli_mask(pn), 25, dev); mft = mask.fraunhofer(193., True)
    eps, N = mask.calculateEpsilonN(mask.deltaK, 25, 193.)
    ab = torch.tensor([0, 0, 0.01, 0, 100, 0.01, 0, 0.01, 0.01, 0.01], dtype=torch.float16)
    pf = L.Pupil(pn, 193., 0.7, ab, dev).generatePupilFunction()
    sh = L.sourceShifts(L.LightSource(0.4, 0.8, pn, 0.7, device=dev).generateQuasar(4, -math.pi / 8), pn)
    sel = sh[(torch.arange(K, device=dev) * sh.shape[0]) // K].contiguous()
    combos = [c.split(",") for c in sys.argv[4:]]
    ref = None
    for batch, xchunk, groups in combos:
        os.environ["LITHO_ABBE_BATCH"] = batch; os.environ["LITHO_ABBE_XCHUNK"] = xchunk; os.environ["LITHO_ABBE_GROUPS"] = groups
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.time()
            out = L.abbeIntensity(mft, pf, sel, N)
            torch.cuda.synchronize(); best = min(best, time.time() - t0)
        if ref is None: ref = out
        err = float((out - ref).abs().max() / ref.max())
        print(f"  batch={batch:>3} xchunk={xchunk:>2} groups={groups} : {best / K * 1e6:6.2f} us/pt  ({K * pn * pn / best:.3e} pt*px/s) relerr_vs_first={err:.1e}", flush=True)
    sys.exit(0)
pn = int(sys.argv[1]); K = int(sys.argv[2])
libs = sys.argv[3].split(",")
combos = sys.argv[4:]
for lib in libs:
    env = dict(os.environ)
    if lib != "default": env["LITHO_ABBE_LIB"] = os.path.join(ROOT, "build", "variants", f"lib_{lib}.so")
    print(f"== lib {lib} pn={pn} K={K}", flush=True)
    subprocess.run([sys.executable, __file__, "--child", str(pn), str(K)] + combos, env=env)
