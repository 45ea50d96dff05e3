#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing the REAL reference
(quarterwave0/LithographySimulator, mounted read-only at /root/reference) in the build
container and running it on CPU.  Only the resulting data (.npz) is committed; the
reference's source never enters this repository and never travels to the GPU box.

    python tests/golden/make_golden.py            # regenerates every fixture

Environment the committed fixtures were made with: torch 2.10.0+rocm7.0 (CPU), 8 threads.
Fixture groups follow SURVEY.md section 8c (G1..G7).
"""
import hashlib
import math
import os
import sys
import zlib

sys.dont_write_bytecode = True          # /root/reference is read-only
REF = os.environ.get("LITHO_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import contextlib
import io

import numpy as np
import torch

import imageformation as ref_if      # noqa: E402  (the reference)
import lightsource as ref_ls         # noqa: E402
import mask as ref_mask              # noqa: E402
import pupil as ref_pupil            # noqa: E402

from lithographysimulator_amd.synthetic import bernoulli_mask, lines_mask  # noqa: E402

ref_if.Mask = ref_mask.Mask          # quirk Q1: abbeImage needs the global name
CPU = torch.device("cpu")
WL, NA, PS = 193.0, 0.7, 25
DEMO_AB = [0, 0, 0.01, 0, 100, 0.01, 0, 0.01, 0.01, 0.01]     # imageformation.py:100
QUASAR = (4, -math.pi / 8)                                      # imageformation.py:112


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def f16(v):
    return torch.tensor(v, dtype=torch.float16)


def source(kind, pn, sin, sout, sx=0.0, sy=0.0, count=4, rot=-math.pi / 8):
    ls = quiet(ref_ls.LightSource, sin, sout, pn, NA, sx, sy, CPU)
    return ls.generateAnnular() if kind == "annular" else ls.generateQuasar(count, rot)


def pupil_fn(pn, ab):
    t = None if ab is None else f16(ab)          # fresh tensor each call (quirk Q2)
    return quiet(ref_pupil.Pupil, pn, WL, NA, t, CPU).generatePupilFunction()


def wavefront(pn, ab):
    t = f16([0]) if ab is None else f16(ab)
    return ref_pupil.generateWavefrontError(t, pn, NA, WL, CPU).real.to(torch.float16)


def shifts_of(bitmap, pn):
    return (torch.argwhere(bitmap) - pn // 2).to(torch.int32)


def raw_image(maskFT, pf, bitmap, N):
    """imageformation.py:54-67 replayed with the reference's own functions, stopping before
    the post-process, so kernel output and post-process are pinned separately (G7)."""
    pn = maskFT.shape[0]
    image = torch.zeros((pn, pn), dtype=torch.complex64)
    sh = shifts_of(bitmap, pn)
    for i in range(sh.shape[0]):
        rolled = torch.roll(pf, shifts=(sh[i, 0], sh[i, 1]), dims=(0, 1))
        image += torch.abs(ref_if.calculateFFTAerial(rolled, maskFT, pn, N)) ** 2
    return image.real.clone()


def subsample_bitmap(bitmap, K):
    pts = torch.argwhere(bitmap)
    S = pts.shape[0]
    idx = (torch.arange(K) * S) // K
    out = torch.zeros_like(bitmap)
    out[pts[idx, 0], pts[idx, 1]] = 1
    return out


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"  wrote {name}  ({os.path.getsize(path) / 1024:.0f} KiB)")


SOURCE_CASES = {
    "circ": dict(kind="annular", sin=0.0, sout=0.5),
    "annular": dict(kind="annular", sin=0.4, sout=0.8),
    "quasar": dict(kind="quasar", sin=0.4, sout=0.8),
    "annular_shift": dict(kind="annular", sin=0.4, sout=0.8, sx=0.25, sy=-0.5),
    "quasar3_oddshift": dict(kind="quasar", sin=0.3, sout=0.9, sx=0.2, sy=-0.1, count=3, rot=0.3),
}


def g1_sources():
    print("G1 sources")
    out = {}
    for pn in (64, 256):
        for name, kw in SOURCE_CASES.items():
            out[f"shifts_{name}_{pn}"] = shifts_of(source(pn=pn, **kw), pn)
    for pn in (1024, 2048, 4096):
        for name, kw in SOURCE_CASES.items():
            if pn == 4096 and name not in ("annular", "quasar", "circ"):
                continue
            bm = source(pn=pn, **kw).numpy().astype(np.uint8)
            packed = np.packbits(bm)
            out[f"count_{name}_{pn}"] = np.int64(bm.sum())
            out[f"sha256_{name}_{pn}"] = np.frombuffer(hashlib.sha256(packed.tobytes()).digest(), dtype=np.uint8)
            if pn <= 2048:
                out[f"packed_{name}_{pn}"] = np.frombuffer(zlib.compress(packed.tobytes(), 9), dtype=np.uint8)
    save("g1_sources.npz", **out)


PUPIL_CASES = {
    "ideal": None,
    "defocus_p100": [0, 0, 0, 0, 100],
    "defocus_m200": [0, 0, 0, 0, -200],
    "defocus_p30": [0, 0, 0, 0, 30],
    "demo": DEMO_AB,
    "short3": [0.1, 0.2, 0.05],
    "terms15": [0, 0, 0, 1, 3, 0, 0, 1, 0, 0, 0.02, 0.03, 0.01, 0.5, 0.2],
}


def g2_pupils():
    print("G2 pupils")
    out = {}
    for pn in (64, 256):
        for name, ab in PUPIL_CASES.items():
            out[f"W_{name}_{pn}"] = wavefront(pn, ab).view(torch.int16)      # fp16 bit patterns
            out[f"phi_{name}_{pn}"] = pupil_fn(pn, ab)
    for pn in (1024, 2048):
        for name in ("ideal", "defocus_p100", "demo"):
            W = wavefront(pn, PUPIL_CASES[name])
            phi = pupil_fn(pn, PUPIL_CASES[name])
            out[f"Wsha_{name}_{pn}"] = np.frombuffer(hashlib.sha256(W.numpy().tobytes()).digest(), dtype=np.uint8)
            out[f"Wsub_{name}_{pn}"] = W[::16, ::16].contiguous().view(torch.int16)
            out[f"phisub_{name}_{pn}"] = phi[::16, ::16].contiguous()
            out[f"nz_{name}_{pn}"] = np.int64((phi != 0).sum())
            out[f"phisum_{name}_{pn}"] = phi.to(torch.complex128).sum().numpy()
    save("g2_pupils.npz", **out)


def mask_of(kind, pn):
    if kind == "bern":
        return bernoulli_mask(pn)
    if kind == "lines":
        return lines_mask(pn)
    return None                                   # the reference's built-in 64x64 demo


def g3_mask_spectra():
    print("G3 mask spectra")
    out = {}
    for pn, ps, kinds in ((64, 25, ("demo", "bern", "lines")), (256, 25, ("bern", "lines")),
                          (64, 48, ("bern",)), (64, 10, ("bern",)), (128, 25, ("bern",)),
                          (96, 25, ("bern",))):
        for kind in kinds:
            mk = quiet(ref_mask.Mask, mask_of(kind, pn), ps, CPU)
            eps, N = mk.calculateEpsilonN(mk.deltaK, ps, WL)
            out[f"spec_{kind}_{pn}_ps{ps}"] = mk.fraunhofer(WL, True)
            out[f"epsN_{kind}_{pn}_ps{ps}"] = np.array([eps, N], dtype=np.float64)
    sizing = []
    for pn in (64, 96, 100, 128, 256, 512, 1024, 2048, 4096, 8192):
        for ps in (5, 10, 25, 48, 64, 65, 100):
            mk = quiet(ref_mask.Mask, torch.zeros(2, 2), ps, CPU)
            eps, N = mk.calculateEpsilonN(4 / pn, ps, WL)
            sizing.append([pn, ps, eps, N])
    out["sizing_table"] = np.array(sizing, dtype=np.float64)
    save("g3_mask_spectra.npz", **out)


def g4_fields():
    print("G4 single-point fields")
    out = {}
    cases = [
        ("demo64", None, 64, 25, DEMO_AB, [(0, 0), (5, -7), (-12, 12), (12, 0), (0, -12), (-3, -9), (25, -30)]),
        ("bern256", "bern", 256, 25, None, [(0, 0), (51, -51), (-40, 13), (100, 90)]),
        ("bern64_Neqpn", "bern", 64, 48, DEMO_AB, [(0, 0), (7, -11)]),
        ("bern64_N4pn", "bern", 64, 10, DEMO_AB, [(0, 0), (-9, 4)]),
        ("bern96", "bern", 96, 25, [0, 0, 0, 0, 50], [(0, 0), (10, -17)]),
    ]
    for tag, kind, pn, ps, ab, shifts in cases:
        mk = quiet(ref_mask.Mask, mask_of(kind, pn) if kind else None, ps, CPU)
        eps, N = mk.calculateEpsilonN(mk.deltaK, ps, WL)
        mft = mk.fraunhofer(WL, True)
        pf = pupil_fn(pn, ab)
        out[f"{tag}_maskFT"] = mft
        out[f"{tag}_pupil"] = pf
        out[f"{tag}_N"] = np.int64(N)
        out[f"{tag}_shifts"] = np.array(shifts, dtype=np.int32)
        out[f"{tag}_fields"] = torch.stack([
            ref_if.calculateFFTAerial(torch.roll(pf, shifts=s, dims=(0, 1)), mft, pn, N) for s in shifts])
    save("g4_fields.npz", **out)


def crop_stats(prefix, img, out, crop=128):
    pn = img.shape[0]
    c0 = pn // 2 - crop // 2
    out[f"{prefix}_crop"] = img[c0:c0 + crop, c0:c0 + crop].contiguous()
    out[f"{prefix}_rowsum"] = img.double().sum(1)
    out[f"{prefix}_colsum"] = img.double().sum(0)
    out[f"{prefix}_max"] = np.float64(img.max())
    out[f"{prefix}_sum"] = np.float64(img.double().sum())
    out[f"{prefix}_shape"] = np.array(img.shape, dtype=np.int64)


def g5_images():
    print("G5/G7 images")
    out = {}
    # demo 64^2, imageformation.py __main__ parameters
    mk = quiet(ref_mask.Mask, None, PS, CPU)
    mft = mk.fraunhofer(WL, True)
    eps, N = mk.calculateEpsilonN(mk.deltaK, PS, WL)
    bm = source("quasar", 64, 0.4, 0.8)
    pf = pupil_fn(64, DEMO_AB)
    out["demo64_final"] = ref_if.abbeImage(mk, mft, pf, bm, PS, mk.deltaK, WL, True, CPU)
    out["demo64_raw"] = raw_image(mft, pf, bm, N)
    # config 1: 256^2, circular sigma 0.5, ideal pupil, full source
    for kind in ("bern", "lines"):
        mk = quiet(ref_mask.Mask, mask_of(kind, 256), PS, CPU)
        mft = mk.fraunhofer(WL, True)
        eps, N = mk.calculateEpsilonN(mk.deltaK, PS, WL)
        bm = source("annular", 256, 0.0, 0.5)
        pf = pupil_fn(256, None)
        out[f"cfg1_{kind}_final"] = ref_if.abbeImage(mk, mft, pf, bm, PS, mk.deltaK, WL, True, CPU)
        out[f"cfg1_{kind}_raw"] = raw_image(mft, pf, bm, N)
        print(f"   cfg1 {kind}: S={int(bm.sum())} sum={float(out[f'cfg1_{kind}_final'].sum()):.6e}")
    # N == pn (pixelSize 48, eps < 1) and N == 4 pn (pixelSize 10) at 64^2, full annular source
    for ps in (48, 10, 64):
        mk = quiet(ref_mask.Mask, bernoulli_mask(64), ps, CPU)
        mft = mk.fraunhofer(WL, True)
        eps, N = mk.calculateEpsilonN(mk.deltaK, ps, WL)
        bm = source("annular", 64, 0.4, 0.8)
        pf = pupil_fn(64, DEMO_AB)
        out[f"bern64_ps{ps}_final"] = ref_if.abbeImage(mk, mft, pf, bm, ps, mk.deltaK, WL, True, CPU)
        out[f"bern64_ps{ps}_raw"] = raw_image(mft, pf, bm, N)
    # non power-of-two mask size
    mk = quiet(ref_mask.Mask, bernoulli_mask(96), PS, CPU)
    mft = mk.fraunhofer(WL, True)
    eps, N = mk.calculateEpsilonN(mk.deltaK, PS, WL)
    bm = subsample_bitmap(source("annular", 96, 0.4, 0.8), 64)
    pf = pupil_fn(96, [0, 0, 0, 0, 50])
    out["bern96_final"] = ref_if.abbeImage(mk, mft, pf, bm, PS, mk.deltaK, WL, True, CPU)
    out["bern96_raw"] = raw_image(mft, pf, bm, N)
    out["bern96_bitmap"] = bm.to(torch.uint8)
    # subsampled-source runs at the BASELINE sizes
    for pn, K, skind, ab in ((1024, 16, "annular", [0, 0, 0, 0, 100]),
                             (2048, 8, "quasar", DEMO_AB),
                             (4096, 4, "annular", [0, 0, 0, 0, 100])):
        mk = quiet(ref_mask.Mask, bernoulli_mask(pn), PS, CPU)
        mft = mk.fraunhofer(WL, True)
        eps, N = mk.calculateEpsilonN(mk.deltaK, PS, WL)
        bm = subsample_bitmap(source(skind, pn, 0.4, 0.8), K)
        pf = pupil_fn(pn, ab)
        final = ref_if.abbeImage(mk, mft, pf, bm, PS, mk.deltaK, WL, True, CPU)
        raw = raw_image(mft, pf, bm, N)
        crop_stats(f"sub{pn}_final", final, out)
        crop_stats(f"sub{pn}_raw", raw, out)
        out[f"sub{pn}_shifts"] = shifts_of(bm, pn)
        # spectrum pins at this size (the spectrum itself is too large to commit)
        out[f"sub{pn}_maskFT_crop"] = mft[pn // 2 - 32:pn // 2 + 32, pn // 2 - 32:pn // 2 + 32].contiguous()
        out[f"sub{pn}_maskFT_abs_sum"] = np.float64(mft.abs().double().sum())
        print(f"   sub {pn}: K={K} shape={tuple(final.shape)} sum={float(final.double().sum()):.6e}")
    save("g5_images.npz", **out)


def g6_through_focus():
    print("G6 through-focus")
    out = {}
    defocus = [-310 + 20 * k for k in range(32)]
    out["defocus_nm"] = np.array(defocus, dtype=np.float64)
    mk = quiet(ref_mask.Mask, None, PS, CPU)
    mft = mk.fraunhofer(WL, True)
    bm = source("quasar", 64, 0.4, 0.8)
    planes = []
    for d in defocus:
        ab = list(DEMO_AB); ab[4] = d
        planes.append(ref_if.abbeImage(mk, mft, pupil_fn(64, ab), bm, PS, mk.deltaK, WL, True, CPU))
    out["stack64_final"] = torch.stack(planes)
    mk = quiet(ref_mask.Mask, bernoulli_mask(256), PS, CPU)
    mft = mk.fraunhofer(WL, True)
    eps, N = mk.calculateEpsilonN(mk.deltaK, PS, WL)
    bm = subsample_bitmap(source("quasar", 256, 0.4, 0.8), 64)
    planes = []
    for d in defocus[::4]:
        ab = list(DEMO_AB); ab[4] = d
        planes.append(raw_image(mft, pupil_fn(256, ab), bm, N))
    out["stack256_raw"] = torch.stack(planes)
    out["stack256_defocus_nm"] = np.array(defocus[::4], dtype=np.float64)
    out["stack256_bitmap"] = bm.to(torch.uint8)
    save("g6_through_focus.npz", **out)


def g8_large_pupils():
    """Pupil support at the sizes where the fp16 sigma grid gets coarse (4096: exact, 8192: the step 4/8192 is
    below fp16 resolution near 1): count, bounding box, hash of W, and a strided sample of phi."""
    print("G8 large pupils")
    out = {}
    for pn in (4096, 8192):
        for name in ("ideal", "defocus_p100"):
            W = wavefront(pn, PUPIL_CASES[name])
            phi = pupil_fn(pn, PUPIL_CASES[name])
            nzmask = phi != 0
            rows = torch.nonzero(nzmask.any(1)).flatten(); cols = torch.nonzero(nzmask.any(0)).flatten()
            out[f"nz_{name}_{pn}"] = np.int64(nzmask.sum())
            out[f"box_{name}_{pn}"] = np.array([int(rows[0]), int(rows[-1]), int(cols[0]), int(cols[-1])], dtype=np.int64)
            out[f"Wsha_{name}_{pn}"] = np.frombuffer(hashlib.sha256(W.numpy().tobytes()).digest(), dtype=np.uint8)
            out[f"phisub_{name}_{pn}"] = phi[::64, ::64].contiguous()
            out[f"rowcount_{name}_{pn}"] = nzmask.sum(1).to(torch.int32)
            del W, phi, nzmask
    save("g8_large_pupils.npz", **out)


def g9_config5_stack():
    """BASELINE config 5 AT ITS SIZE: 2048^2, 32 through-focus planes (aberrations[4] = -310 + 20 k nm, the other
    terms = the demo vector), K = 3 source points strided through the quasar list.  The reference has no stack
    API: this is its Python loop over Pupil(...) + abbeImage(...) (imageformation.py:47-77, pupil.py:91-92).
    Per plane: centre crop, fp64 row / column sums, max and sum of the raw intensity and of the final image."""
    print("G9 config-5 stack (2048^2 x 32 planes)")
    pn, K = 2048, 3
    out = {}
    defocus = [-310 + 20 * k for k in range(32)]
    out["defocus_nm"] = np.array(defocus, dtype=np.float64)
    mk = quiet(ref_mask.Mask, bernoulli_mask(pn), PS, CPU)
    mft = mk.fraunhofer(WL, True)
    eps, N = mk.calculateEpsilonN(mk.deltaK, PS, WL)
    bm = subsample_bitmap(source("quasar", pn, 0.4, 0.8), K)
    out["shifts"] = shifts_of(bm, pn)
    keys = ("crop", "rowsum", "colsum", "max", "sum")
    acc = {f"{kind}_{k}": [] for kind in ("raw", "final") for k in keys}
    for i, d in enumerate(defocus):
        ab = list(DEMO_AB); ab[4] = d
        pf = pupil_fn(pn, ab)
        tmp = {}
        crop_stats("raw", raw_image(mft, pf, bm, N), tmp, crop=64)
        crop_stats("final", ref_if.abbeImage(mk, mft, pf, bm, PS, mk.deltaK, WL, True, CPU), tmp, crop=64)
        for kind in ("raw", "final"):
            for k in keys:
                v = tmp[f"{kind}_{k}"]
                acc[f"{kind}_{k}"].append(v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
        print(f"   plane {i:2d} d={d:5d} nm  raw sum {float(tmp['raw_sum']):.6e}", flush=True)
    for k, v in acc.items():
        out[k] = np.stack(v)
    out["final_shape"] = tmp["final_shape"]
    save("g9_config5_stack.npz", **out)


def g10_contiguous_shards():
    """Long CONSECUTIVE runs of source points at the BASELINE sizes (what one rank of a many-GPU run accumulates):
    config 2 (1024^2, annular, defocus pupil) points [40000, 42048) and config 3 (2048^2, quasar, demo pupil) points
    [100000, 100512) of the row-major list, accumulated by the reference's own loop in its own order (fp32,
    sequential).  A full-S image at these sizes would take the reference 2 h (config 2) to 24 h (config 3) on this
    container's 8 cores, so full-S parity is pinned by these shard-sized runs plus the additivity property."""
    print("G10 contiguous shards")
    out = {}
    for tag, pn, skind, ab, lo, n in (("cfg2", 1024, "annular", [0, 0, 0, 0, 100], 40000, 2048),
                                      ("cfg3", 2048, "quasar", DEMO_AB, 100000, 512)):
        mk = quiet(ref_mask.Mask, bernoulli_mask(pn), PS, CPU)
        mft = mk.fraunhofer(WL, True)
        eps, N = mk.calculateEpsilonN(mk.deltaK, PS, WL)
        full = source(skind, pn, 0.4, 0.8)
        pts = torch.argwhere(full)
        bm = torch.zeros_like(full)
        sel = pts[lo:lo + n]
        bm[sel[:, 0], sel[:, 1]] = 1
        pf = pupil_fn(pn, ab)
        raw = raw_image(mft, pf, bm, N)
        crop_stats(f"{tag}_raw", raw, out)
        out[f"{tag}_range"] = np.array([lo, lo + n, pts.shape[0]], dtype=np.int64)
        out[f"{tag}_first_last_shift"] = shifts_of(bm, pn)[[0, -1]]
        print(f"   {tag}: points [{lo},{lo + n}) of {pts.shape[0]}  sum={float(raw.double().sum()):.6e}", flush=True)
    save("g10_contiguous_shards.npz", **out)


def full_image_with_raw(mk, mft, pf, bm):
    """ONE run of the reference's own abbeImage (imageformation.py:47-77) that also yields the raw accumulated
    intensity: the tensor the reference hands to F.interpolate at :71 is abs(image), i.e. the loop's sum."""
    import torch.nn.functional as F
    captured = {}
    orig = F.interpolate

    def spy(inp, *a, **k):
        captured["raw"] = inp.squeeze(0).squeeze(0).clone()
        return orig(inp, *a, **k)

    F.interpolate = spy
    try:
        final = ref_if.abbeImage(mk, mft, pf, bm, PS, mk.deltaK, WL, True, CPU)
    finally:
        F.interpolate = orig
    return final, captured["raw"]


def g11_config2_full():
    """BASELINE config 2 IN FULL, by the reference itself: 1024^2 bernoulli mask, annular 0.4-0.8 (S = 98,832),
    defocus-only pupil [0,0,0,0,100] -- the reference's sequential fp32 loop over every source point (about two hours
    on this container's CPUs).  Stored: centre crops, a stride-8 sample of every region, fp64 row / column sums, max
    and sum of the raw accumulated intensity and of the final image."""
    print("G11 config 2 in full (reference loop over 98,832 source points)")
    import time
    pn = 1024
    out = {}
    mk = quiet(ref_mask.Mask, bernoulli_mask(pn), PS, CPU)
    mft = mk.fraunhofer(WL, True)
    bm = source("annular", pn, 0.4, 0.8)
    pf = pupil_fn(pn, [0, 0, 0, 0, 100])
    out["S"] = np.int64(bm.sum())
    t0 = time.time()
    final, raw = full_image_with_raw(mk, mft, pf, bm)
    out["reference_seconds"] = np.float64(time.time() - t0)
    out["reference_threads"] = np.int64(torch.get_num_threads())
    for tag, img in (("final", final), ("raw", raw)):
        crop_stats(f"cfg2full_{tag}", img, out)
        out[f"cfg2full_{tag}_stride8"] = img[::8, ::8].contiguous()
    print(f"   S={int(out['S'])}  {float(out['reference_seconds']):.0f} s  final sum {float(final.double().sum()):.7e} "
          f"max {float(final.max()):.7e}", flush=True)
    save("g11_config2_full.npz", **out)


def g12_shard4096():
    """A consecutive 64-point shard at 4096^2 (BASELINE config 4: annular 0.4-0.8, defocus pupil), points
    [800000, 800064) of the row-major list, by the reference's own abbeImage: raw intensity and the 4094^2 final
    image (quirk Q5) over a run that spans several launch batches of the engine."""
    print("G12 consecutive shard at 4096^2")
    pn, lo, n = 4096, 800000, 64
    out = {}
    mk = quiet(ref_mask.Mask, bernoulli_mask(pn), PS, CPU)
    mft = mk.fraunhofer(WL, True)
    full = source("annular", pn, 0.4, 0.8)
    pts = torch.argwhere(full)
    bm = torch.zeros_like(full)
    sel = pts[lo:lo + n]
    bm[sel[:, 0], sel[:, 1]] = 1
    pf = pupil_fn(pn, [0, 0, 0, 0, 100])
    final, raw = full_image_with_raw(mk, mft, pf, bm)
    for tag, img in (("final", final), ("raw", raw)):
        crop_stats(f"cfg4shard_{tag}", img, out)
        out[f"cfg4shard_{tag}_stride32"] = img[::32, ::32].contiguous()
    out["cfg4shard_range"] = np.array([lo, lo + n, pts.shape[0]], dtype=np.int64)
    out["cfg4shard_first_last_shift"] = shifts_of(bm, pn)[[0, -1]]
    print(f"   points [{lo},{lo + n}) of {pts.shape[0]}  final shape {tuple(final.shape)}  sum={float(final.double().sum()):.7e}", flush=True)
    save("g12_shard4096.npz", **out)


def g17_config4_fold_run():
    """4,000 CONSECUTIVE source points of BASELINE config 4 (4096^2 bernoulli mask, annular 0.4-0.8, defocus pupil
    [0,0,0,0,100]), points [400000, 404000) of the row-major list, by the reference's own abbeImage (raw intensity
    captured from the same run): 66 launch batches of the engine's default 60-point batch at this size plus 40 points,
    i.e. MORE than one 64-batch slab fold (3,840 points), so the engine's two-level summation at config 4's default
    launch geometry is compared with the reference's sequential fp32 loop on dense data.  Between two and three hours
    of this container's CPUs (about 2-3 s per point)."""
    print("G17 config 4, 4000 consecutive points at 4096^2")
    import time
    pn, lo, n = 4096, 400000, 4000
    out = {}
    mk = quiet(ref_mask.Mask, bernoulli_mask(pn), PS, CPU)
    mft = mk.fraunhofer(WL, True)
    full = source("annular", pn, 0.4, 0.8)
    pts = torch.argwhere(full)
    bm = torch.zeros_like(full)
    sel = pts[lo:lo + n]
    bm[sel[:, 0], sel[:, 1]] = 1
    pf = pupil_fn(pn, [0, 0, 0, 0, 100])
    t0 = time.time()
    final, raw = full_image_with_raw(mk, mft, pf, bm)
    out["reference_seconds"] = np.float64(time.time() - t0)
    out["reference_threads"] = np.int64(torch.get_num_threads())
    for tag, img in (("final", final), ("raw", raw)):
        crop_stats(f"cfg4run_{tag}", img, out)
        out[f"cfg4run_{tag}_stride32"] = img[::32, ::32].contiguous()
        out[f"cfg4run_{tag}_rows"] = img[[0, 1, 1023, 2047, 2048, 3071, img.shape[0] - 2, img.shape[0] - 1], :].contiguous()
    out["cfg4run_range"] = np.array([lo, lo + n, pts.shape[0]], dtype=np.int64)
    out["cfg4run_first_last_shift"] = shifts_of(bm, pn)[[0, -1]]
    print(f"   points [{lo},{lo + n}) of {pts.shape[0]}  {float(out['reference_seconds']):.0f} s  final shape "
          f"{tuple(final.shape)}  sum={float(final.double().sum()):.7e}", flush=True)
    save("g17_config4_fold_run.npz", **out)


def g13_config3_long_run():
    """1,536 CONSECUTIVE source points of BASELINE config 3 (2048^2 bernoulli mask, quasar 0.4-0.8, 10-term demo pupil),
    points [60000, 61536) of the row-major list, by the reference's own abbeImage (raw intensity captured from the same
    run): 128 launch batches of the engine's default 12-point batch at this size, i.e. MORE than the 64-batch slab fold,
    so the engine's two-level summation at its default launch geometry is compared with the reference's sequential fp32
    loop on dense data.  About ten minutes of this container's CPUs."""
    print("G13 config 3, 1536 consecutive points at 2048^2")
    import time
    pn, lo, n = 2048, 60000, 1536
    out = {}
    mk = quiet(ref_mask.Mask, bernoulli_mask(pn), PS, CPU)
    mft = mk.fraunhofer(WL, True)
    full = source("quasar", pn, 0.4, 0.8)
    pts = torch.argwhere(full)
    bm = torch.zeros_like(full)
    sel = pts[lo:lo + n]
    bm[sel[:, 0], sel[:, 1]] = 1
    pf = pupil_fn(pn, DEMO_AB)
    t0 = time.time()
    final, raw = full_image_with_raw(mk, mft, pf, bm)
    out["reference_seconds"] = np.float64(time.time() - t0)
    for tag, img in (("final", final), ("raw", raw)):
        crop_stats(f"cfg3run_{tag}", img, out)
        out[f"cfg3run_{tag}_stride16"] = img[::16, ::16].contiguous()
    out["cfg3run_range"] = np.array([lo, lo + n, pts.shape[0]], dtype=np.int64)
    out["cfg3run_first_last_shift"] = shifts_of(bm, pn)[[0, -1]]
    print(f"   points [{lo},{lo + n}) of {pts.shape[0]}  {float(out['reference_seconds']):.0f} s  final sum "
          f"{float(final.double().sum()):.7e}", flush=True)
    save("g13_config3_long_run.npz", **out)


def g14_config3_rank_shard():
    """ONE RANK'S SHARD of BASELINE config 3 at 8 GPUs, in full, by the reference itself: source points [0, 24764) of the
    2048^2 quasar list (distributed.shard_bounds(198108, 0, 8)) through the reference's own abbeImage -- its sequential fp32
    sum of 24,764 images (about an hour and a half of this container's CPUs).  The engine runs this as 2,064 of its default
    12-point batches with 32 slab folds: the per-rank work of the 8-GPU run, on dense reference-made data."""
    print("G14 config 3, rank 0 of 8: 24,764 consecutive points at 2048^2")
    import time
    pn, lo, n = 2048, 0, 24764
    out = {}
    mk = quiet(ref_mask.Mask, bernoulli_mask(pn), PS, CPU)
    mft = mk.fraunhofer(WL, True)
    full = source("quasar", pn, 0.4, 0.8)
    pts = torch.argwhere(full)
    assert pts.shape[0] == 198108
    bm = torch.zeros_like(full)
    sel = pts[lo:lo + n]
    bm[sel[:, 0], sel[:, 1]] = 1
    pf = pupil_fn(pn, DEMO_AB)
    t0 = time.time()
    final, raw = full_image_with_raw(mk, mft, pf, bm)
    out["reference_seconds"] = np.float64(time.time() - t0)
    out["reference_threads"] = np.int64(torch.get_num_threads())
    for tag, img in (("final", final), ("raw", raw)):
        crop_stats(f"cfg3shard_{tag}", img, out)
        out[f"cfg3shard_{tag}_stride16"] = img[::16, ::16].contiguous()
    out["cfg3shard_range"] = np.array([lo, lo + n, pts.shape[0]], dtype=np.int64)
    out["cfg3shard_first_last_shift"] = shifts_of(bm, pn)[[0, -1]]
    print(f"   points [{lo},{lo + n}) of {pts.shape[0]}  {float(out['reference_seconds']):.0f} s  final sum "
          f"{float(final.double().sum()):.7e}", flush=True)
    save("g14_config3_rank_shard.npz", **out)


def g15_config5_stack_run():
    """BASELINE config 5's stack on DENSE data through several launch batches per plane: 8 of its 32 planes (defocus -310 + 80 j
    nm, j = 0..7) x 240 CONSECUTIVE source points [90000, 90240) of the 2048^2 quasar list, by the reference's Python loop over
    Pupil(...) + abbeImage(...) (there is no stack API in the reference; imageformation.py:47-77, pupil.py:91-92): 20 of the
    engine's default 12-point batches per plane.  Per plane: centre crop, stride-32 grid, fp64 row / column sums, max and sum
    of the raw intensity and of the final image.  About ten minutes of this container's CPUs."""
    print("G15 config-5 stack, 8 planes x 240 consecutive points at 2048^2")
    pn, lo, n = 2048, 90000, 240
    out = {}
    defocus = [-310 + 80 * j for j in range(8)]
    out["defocus_nm"] = np.array(defocus, dtype=np.float64)
    mk = quiet(ref_mask.Mask, bernoulli_mask(pn), PS, CPU)
    mft = mk.fraunhofer(WL, True)
    full = source("quasar", pn, 0.4, 0.8)
    pts = torch.argwhere(full)
    bm = torch.zeros_like(full)
    sel = pts[lo:lo + n]
    bm[sel[:, 0], sel[:, 1]] = 1
    out["range"] = np.array([lo, lo + n, pts.shape[0]], dtype=np.int64)
    out["first_last_shift"] = shifts_of(bm, pn)[[0, -1]]
    keys = ("crop", "rowsum", "colsum", "max", "sum", "stride32")
    acc = {f"{kind}_{k}": [] for kind in ("raw", "final") for k in keys}
    for i, d in enumerate(defocus):
        ab = list(DEMO_AB); ab[4] = d
        pf = pupil_fn(pn, ab)
        final, raw = full_image_with_raw(mk, mft, pf, bm)
        tmp = {}
        for kind, img in (("raw", raw), ("final", final)):
            crop_stats(kind, img, tmp, crop=64)
            tmp[f"{kind}_stride32"] = img[::32, ::32].contiguous()
            for k in keys:
                v = tmp[f"{kind}_{k}"]
                acc[f"{kind}_{k}"].append(v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
        print(f"   plane {i} d={d:5d} nm  raw sum {float(tmp['raw_sum']):.6e}", flush=True)
    for k, v in acc.items():
        out[k] = np.stack(v)
    out["final_shape"] = tmp["final_shape"]
    save("g15_config5_stack_run.npz", **out)


def odd_mask(pn):
    """Deterministic 0/1 mask for sizes that are not multiples of 64 (integer hash, as synthetic.bernoulli_mask)."""
    return bernoulli_mask(pn)


def g16_odd_sizes():
    """Mask sizes that are NOT powers of two, by the reference itself (its torch.fft path takes any size): 200^2 (N 512),
    1000^2 (N 2048), 1500^2 (N 2048) with the demo pupil and annular 0.4-0.8 sources -- K strided points each, plus 600
    CONSECUTIVE points at 1000^2 (a source list long enough for the engine's coarse grid at the padded size).  The engine runs
    these embedded in 256^2 / 1024^2 / 2048^2 grids; the reference knows nothing of that."""
    print("G16 odd mask sizes")
    out = {}
    for tag, pn, K in (("p200", 200, 24), ("p1000", 1000, 12), ("p1500", 1500, 6)):
        mk = quiet(ref_mask.Mask, odd_mask(pn), PS, CPU)
        mft = mk.fraunhofer(WL, True)
        eps, N = mk.calculateEpsilonN(mk.deltaK, PS, WL)
        full = source("annular", pn, 0.4, 0.8)
        bm = subsample_bitmap(full, K)
        pf = pupil_fn(pn, DEMO_AB)
        final, raw = full_image_with_raw(mk, mft, pf, bm)
        out[f"{tag}_N"] = np.int64(N)
        out[f"{tag}_S_full"] = np.int64(full.sum())
        out[f"{tag}_shifts"] = shifts_of(bm, pn)
        for kind, img in (("raw", raw), ("final", final)):
            crop_stats(f"{tag}_{kind}", img, out, crop=min(128, pn))
            if pn <= 200:
                out[f"{tag}_{kind}_image"] = img.contiguous()
        print(f"   {tag}: N {N}  K {K}  final {tuple(final.shape)}  sum {float(final.double().sum()):.7e}", flush=True)
    # a consecutive run at 1000^2
    pn, lo, n = 1000, 30000, 600
    mk = quiet(ref_mask.Mask, odd_mask(pn), PS, CPU)
    mft = mk.fraunhofer(WL, True)
    full = source("annular", pn, 0.4, 0.8)
    pts = torch.argwhere(full)
    bm = torch.zeros_like(full)
    sel = pts[lo:lo + n]
    bm[sel[:, 0], sel[:, 1]] = 1
    final, raw = full_image_with_raw(mk, mft, pupil_fn(pn, DEMO_AB), bm)
    out["run1000_range"] = np.array([lo, lo + n, pts.shape[0]], dtype=np.int64)
    out["run1000_first_last_shift"] = shifts_of(bm, pn)[[0, -1]]
    for kind, img in (("raw", raw), ("final", final)):
        crop_stats(f"run1000_{kind}", img, out)
        out[f"run1000_{kind}_stride8"] = img[::8, ::8].contiguous()
    print(f"   run1000: points [{lo},{lo + n}) of {pts.shape[0]}  final sum {float(final.double().sum()):.7e}", flush=True)
    save("g16_odd_sizes.npz", **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    if os.environ.get("LITHO_GOLDEN_THREADS"):
        torch.set_num_threads(int(os.environ["LITHO_GOLDEN_THREADS"]))
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g8", "g9", "g10"]
    for g in which:
        {"g1": g1_sources, "g2": g2_pupils, "g3": g3_mask_spectra, "g4": g4_fields,
         "g5": g5_images, "g6": g6_through_focus, "g8": g8_large_pupils, "g9": g9_config5_stack, "g10": g10_contiguous_shards,
         "g11": g11_config2_full, "g12": g12_shard4096, "g13": g13_config3_long_run, "g14": g14_config3_rank_shard, "g15": g15_config5_stack_run, "g16": g16_odd_sizes, "g17": g17_config4_fold_run}[g]()
