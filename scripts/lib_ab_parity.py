"""Image of one variant library against the product library on the same inputs (two fresh processes):
    python scripts/lib_ab_parity.py pn K variant[,variant2...]      (build/variants/lib_<variant>.so)
Prints max|diff| / max and the kernels each run launched; config-3 optics on the bernoulli mask, K strided source points."""
import math, os, subprocess, sys
os.environ.setdefault("LITHO_ABBE_COARSE", "2")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    import lithographysimulator_amd as L
    from lithographysimulator_amd import _native as nat
    from lithographysimulator_amd.synthetic import bernoulli_mask
    pn, K, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    dev = torch.device("cuda", 0)
    mask = L.Mask(bernoulli_mask(pn), 25, dev); mft = mask.fraunhofer(193., True)
    eps, N = mask.calculateEpsilonN(mask.deltaK, 25, 193.)
    pf = L.Pupil(pn, 193., 0.7, torch.tensor([0, 0, 0.01, 0, 100, 0.01, 0, 0.01, 0.01, 0.01], dtype=torch.float16), dev).generatePupilFunction()
    sh = L.sourceShifts(L.LightSource(0.4, 0.8, pn, 0.7, device=dev).generateAnnular(), pn)
    sel = sh[(torch.arange(K, device=dev) * sh.shape[0]) // K].contiguous()
    img = L.abbeIntensity(mft, pf, sel, N); torch.cuda.synchronize()
    print("   kernels", nat.last_kernels(), "plan batch", nat.last_plan()["batch"], flush=True)
    np.save(out, img.cpu().numpy())
    sys.exit(0)
import numpy as np
pn, K = sys.argv[1], sys.argv[2]
imgs = {}
for lib in ["default"] + sys.argv[3].split(","):
    env = dict(os.environ, LITHO_ALLOW_DIAG="1")
    if lib != "default": env["LITHO_ABBE_LIB"] = os.path.join(ROOT, "build", "variants", f"lib_{lib}.so")
    out = f"/tmp/lib_ab_{lib}.npy"
    print(f"== {lib}", flush=True)
    subprocess.run([sys.executable, __file__, "--child", pn, K, out], env=env, check=True)
    imgs[lib] = np.load(out).astype(np.float64)
ref = imgs["default"]
for lib, im in imgs.items():
    if lib == "default": continue
    print(f"{lib} vs default: max|diff| / max = {np.abs(im - ref).max() / ref.max():.3e}   (nan: {int(np.isnan(im).sum())})", flush=True)
