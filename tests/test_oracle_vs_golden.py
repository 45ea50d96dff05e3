"""Pins the CPU oracle (oracle/abbe_oracle.py, oracle/abbe_ref.c) against golden vectors
captured from the real reference (tests/golden/make_golden.py).  CPU only."""
import math

import numpy as np
import pytest
import torch

from oracle import abbe_oracle as O
from lithographysimulator_amd.synthetic import bernoulli_mask, lines_mask
from helpers import (DEMO_AB, NA, PS, PUPIL_CASES, SOURCE_CASES, TOL_FIELD, TOL_IMAGE_L2, TOL_IMAGE_MAX, WL,
                     c_oracle_field, crop_center, f16, load_c_oracle, rel_l2, rel_max, sha256_packed,
                     subsample_bitmap, unpack_bitmap)


def make_source(pn, kind, sin, sout, sx=0.0, sy=0.0, count=4, rot=-math.pi / 8):
    if kind == "annular":
        return O.source_annular(sin, sout, pn, sx, sy)
    return O.source_quasar(sin, sout, pn, count, rot, sx, sy)


# ---------------------------------------------------------------- G1
@pytest.mark.parametrize("pn", [64, 256])
@pytest.mark.parametrize("name", list(SOURCE_CASES))
def test_source_lists_exact(golden, pn, name):
    g = golden("g1_sources.npz")
    got = O.source_shifts(make_source(pn, **SOURCE_CASES[name]), pn).numpy()
    assert np.array_equal(got, g[f"shifts_{name}_{pn}"])


@pytest.mark.parametrize("pn", [1024, 2048])
@pytest.mark.parametrize("name", list(SOURCE_CASES))
def test_source_bitmaps_large_exact(golden, pn, name):
    g = golden("g1_sources.npz")
    bm = make_source(pn, **SOURCE_CASES[name]).numpy()
    assert int(bm.sum()) == int(g[f"count_{name}_{pn}"])
    assert np.array_equal(sha256_packed(bm), g[f"sha256_{name}_{pn}"])
    assert np.array_equal(bm, unpack_bitmap(g[f"packed_{name}_{pn}"], pn))


def test_source_bitmap_4096_counts(golden):
    g = golden("g1_sources.npz")
    bm = make_source(4096, **SOURCE_CASES["annular"]).numpy()
    assert int(bm.sum()) == int(g["count_annular_4096"]) == 1581616
    assert np.array_equal(sha256_packed(bm), g["sha256_annular_4096"])


# ---------------------------------------------------------------- G2
@pytest.mark.parametrize("pn", [64, 256])
@pytest.mark.parametrize("name", list(PUPIL_CASES))
def test_pupil_exact(golden, pn, name):
    g = golden("g2_pupils.npz")
    ab = PUPIL_CASES[name]
    W = O.wavefront_error(f16([0]) if ab is None else f16(ab), pn, NA, WL)
    assert np.array_equal(W.view(torch.int16).numpy(), g[f"W_{name}_{pn}"])
    phi = O.pupil_function(None if ab is None else f16(ab), pn, NA, WL)
    assert np.array_equal(phi.numpy(), g[f"phi_{name}_{pn}"])


@pytest.mark.parametrize("pn", [1024, 2048])
@pytest.mark.parametrize("name", ["ideal", "defocus_p100", "demo"])
def test_pupil_large_exact(golden, pn, name):
    import hashlib
    g = golden("g2_pupils.npz")
    ab = PUPIL_CASES[name]
    W = O.wavefront_error(f16([0]) if ab is None else f16(ab), pn, NA, WL)
    assert np.array_equal(np.frombuffer(hashlib.sha256(W.numpy().tobytes()).digest(), dtype=np.uint8),
                          g[f"Wsha_{name}_{pn}"])
    phi = O.pupil_from_wavefront(W, pn)
    assert np.array_equal(phi[::16, ::16].numpy(), g[f"phisub_{name}_{pn}"])
    assert int((phi != 0).sum()) == int(g[f"nz_{name}_{pn}"])


def test_pupil_length4_raises_like_reference():
    with pytest.raises(IndexError):
        O.wavefront_error(f16([0, 0, 0, 1]), 64, NA, WL)


def test_osa_indices():
    assert [O.osa_index_to_mn(j) for j in range(10)] == [
        (0, 0), (-1, 1), (1, 1), (-2, 2), (0, 2), (2, 2), (-3, 3), (-1, 3), (1, 3), (3, 3)]


# ---------------------------------------------------------------- G3 + sizing
def test_sizing_table(golden):
    for pn, ps, eps, N in golden("g3_mask_spectra.npz")["sizing_table"]:
        e, n = O.calculate_epsilon_n(4 / pn, ps, WL)
        assert n == int(N) and e == eps


@pytest.mark.parametrize("key", ["demo_64_ps25", "bern_64_ps25", "lines_64_ps25", "bern_256_ps25", "lines_256_ps25",
                                 "bern_64_ps48", "bern_64_ps10", "bern_128_ps25", "bern_96_ps25"])
def test_mask_spectrum(golden, key):
    g = golden("g3_mask_spectra.npz")
    kind, pn, ps = key.split("_"); pn = int(pn); ps = int(ps[2:])
    if kind == "demo":
        geo = torch.zeros((64, 64), dtype=torch.int16)
        for c0 in (16, 25, 34, 43):
            geo[9:55, c0:c0 + 4] = 1
        assert torch.equal(geo, lines_mask(64))
    else:
        geo = bernoulli_mask(pn) if kind == "bern" else lines_mask(pn)
    got = O.mask_spectrum(geo, ps, WL)
    assert rel_max(got, torch.from_numpy(g[f"spec_{key}"])) < 1e-6


# ---------------------------------------------------------------- G4
@pytest.mark.parametrize("tag", ["demo64", "bern256", "bern64_Neqpn", "bern64_N4pn", "bern96"])
def test_fields(golden, tag):
    g = golden("g4_fields.npz")
    mft = torch.from_numpy(g[f"{tag}_maskFT"]); pf = torch.from_numpy(g[f"{tag}_pupil"])
    N = int(g[f"{tag}_N"]); pn = mft.shape[0]
    lib = load_c_oracle() if pn <= 96 else None
    for s, ref in zip(g[f"{tag}_shifts"], g[f"{tag}_fields"]):
        ref = torch.from_numpy(ref)
        chain = O.field_opchain(torch.roll(pf, shifts=(int(s[0]), int(s[1])), dims=(0, 1)), mft, pn, N)
        assert rel_max(chain, ref) < 1e-6
        closed = O.field_closed_form(pf, mft, int(s[0]), int(s[1]), N)
        assert rel_max(closed, ref) < TOL_FIELD
        if lib is not None:
            assert rel_max(c_oracle_field(lib, pf, mft, N, s[0], s[1]), ref) < TOL_FIELD


# ---------------------------------------------------------------- G5 / G7
def _demo_inputs():
    geo = lines_mask(64)
    mft = O.mask_spectrum(geo, PS, WL)
    bm = O.source_quasar(0.4, 0.8, 64, 4, -math.pi / 8)
    pf = O.pupil_function(f16(DEMO_AB), 64, NA, WL)
    return mft, bm, pf


def test_demo_image(golden):
    g = golden("g5_images.npz")
    mft, bm, pf = _demo_inputs()
    eps, N = O.calculate_epsilon_n(4 / 64, PS, WL)
    raw = O.abbe_raw(mft, pf, O.source_shifts(bm, 64), N)
    assert rel_max(raw, g["demo64_raw"]) < TOL_IMAGE_MAX and rel_l2(raw, g["demo64_raw"]) < TOL_IMAGE_L2
    final = O.post_process(raw, eps)
    assert final.shape == g["demo64_final"].shape
    assert rel_max(final, g["demo64_final"]) < TOL_IMAGE_MAX
    assert abs(float(final.sum()) / 2.2029254e13 - 1) < 1e-5      # SURVEY 3.1 [ran]
    raw64 = O.abbe_raw_f64(mft, pf, O.source_shifts(bm, 64), N)
    assert rel_max(raw64, g["demo64_raw"]) < TOL_IMAGE_MAX


def test_demo_image_c_oracle(golden):
    g = golden("g5_images.npz")
    mft, bm, pf = _demo_inputs()
    eps, N = O.calculate_epsilon_n(4 / 64, PS, WL)
    sh = np.ascontiguousarray(O.source_shifts(bm, 64).numpy())
    lib = load_c_oracle()
    img = np.zeros((64, 64), dtype=np.float64)
    p = np.ascontiguousarray(pf.numpy()); m = np.ascontiguousarray(mft.numpy())
    assert lib.oracle_abbe_accumulate(p.ctypes.data, m.ctypes.data, sh.ctypes.data, sh.shape[0], 64, N,
                                      img.ctypes.data) == 0
    assert rel_max(torch.from_numpy(img), g["demo64_raw"]) < TOL_IMAGE_MAX


@pytest.mark.parametrize("kind", ["bern", "lines"])
def test_config1_full(golden, kind):
    """BASELINE config 1: 256^2, circular sigma 0.5, ideal pupil, all 3233 source points."""
    g = golden("g5_images.npz")
    geo = bernoulli_mask(256) if kind == "bern" else lines_mask(256)
    mft = O.mask_spectrum(geo, PS, WL)
    bm = O.source_annular(0.0, 0.5, 256)
    pf = O.pupil_function(None, 256, NA, WL)
    eps, N = O.calculate_epsilon_n(4 / 256, PS, WL)
    raw = O.abbe_raw(mft, pf, O.source_shifts(bm, 256), N)
    assert rel_max(raw, g[f"cfg1_{kind}_raw"]) < TOL_IMAGE_MAX and rel_l2(raw, g[f"cfg1_{kind}_raw"]) < TOL_IMAGE_L2
    final = O.post_process(raw, eps)
    assert rel_max(final, g[f"cfg1_{kind}_final"]) < TOL_IMAGE_MAX


@pytest.mark.parametrize("ps", [48, 10])
def test_other_fft_sizes(golden, ps):
    g = golden("g5_images.npz")
    mft = O.mask_spectrum(bernoulli_mask(64), ps, WL)
    bm = O.source_annular(0.4, 0.8, 64)
    pf = O.pupil_function(f16(DEMO_AB), 64, NA, WL)
    final = O.abbe_image(mft, pf, bm, ps, 4 / 64, WL)
    assert final.shape == g[f"bern64_ps{ps}_final"].shape
    assert rel_max(final, g[f"bern64_ps{ps}_final"]) < TOL_IMAGE_MAX


def test_odd_mask_size_200(golden):
    """A mask size that is not a power of two, by the reference (golden g16): 200^2, N = 512, 24 strided points of the annular
    source, the whole raw and final image -- the final one is 198 x 198 (the reference's pad arithmetic, as 4096 -> 4094)."""
    g = golden("g16_odd_sizes.npz")
    pn = 200
    mft = O.mask_spectrum(bernoulli_mask(pn), PS, WL)
    eps, N = O.calculate_epsilon_n(4 / pn, PS, WL)
    assert N == int(g["p200_N"]) == 512
    bm = O.source_annular(0.4, 0.8, pn)
    assert int(bm.sum()) == int(g["p200_S_full"])
    sh = torch.from_numpy(g["p200_shifts"])
    pf = O.pupil_function(f16(DEMO_AB), pn, NA, WL)
    raw = O.abbe_raw(mft, pf, sh, N)
    assert rel_max(raw, g["p200_raw_image"]) < TOL_IMAGE_MAX and rel_l2(raw, g["p200_raw_image"]) < TOL_IMAGE_L2
    final = O.post_process(raw, eps)
    assert tuple(final.shape) == (198, 198) == tuple(g["p200_final_shape"])
    assert rel_max(final, g["p200_final_image"]) < TOL_IMAGE_MAX


def test_n_smaller_than_mask_raises():
    mft = O.mask_spectrum(bernoulli_mask(64), 25, WL)
    with pytest.raises(RuntimeError):
        O.abbe_raw(mft, mft, torch.zeros((1, 2), dtype=torch.int32), 32)


def test_subsampled_1024(golden):
    g = golden("g5_images.npz")
    pn = 1024
    mft = O.mask_spectrum(bernoulli_mask(pn), PS, WL)
    assert rel_max(mft[pn // 2 - 32:pn // 2 + 32, pn // 2 - 32:pn // 2 + 32], g["sub1024_maskFT_crop"]) < 1e-6
    bm = subsample_bitmap(O.source_annular(0.4, 0.8, pn), 16)
    sh = O.source_shifts(bm, pn)
    assert np.array_equal(sh.numpy(), g["sub1024_shifts"])
    pf = O.pupil_function(f16([0, 0, 0, 0, 100]), pn, NA, WL)
    eps, N = O.calculate_epsilon_n(4 / pn, PS, WL)
    raw = O.abbe_raw(mft, pf, sh, N)
    assert rel_max(crop_center(raw), g["sub1024_raw_crop"]) < TOL_IMAGE_MAX
    assert np.allclose(raw.double().sum(1).numpy(), g["sub1024_raw_rowsum"], rtol=1e-5)
    final = O.post_process(raw, eps)
    assert tuple(final.shape) == tuple(g["sub1024_final_shape"])
    assert rel_max(crop_center(final), g["sub1024_final_crop"]) < TOL_IMAGE_MAX


def test_post_process_4096_shape():
    eps, N = O.calculate_epsilon_n(4 / 4096, PS, WL)
    out = O.post_process(torch.ones(4096, 4096), eps)
    assert out.shape == (4094, 4094)                     # quirk Q5


# ---------------------------------------------------------------- G6
def test_through_focus_64(golden):
    g = golden("g6_through_focus.npz")
    geo = lines_mask(64)
    mft = O.mask_spectrum(geo, PS, WL)
    bm = O.source_quasar(0.4, 0.8, 64, 4, -math.pi / 8)
    for k in (0, 13, 31):
        ab = list(DEMO_AB); ab[4] = float(g["defocus_nm"][k])
        pf = O.pupil_function(f16(ab), 64, NA, WL)
        final = O.abbe_image(mft, pf, bm, PS, 4 / 64, WL)
        assert rel_max(final, g["stack64_final"][k]) < TOL_IMAGE_MAX


# ---------------------------------------------------------------- G9 (config 5 at its size)
@pytest.mark.parametrize("k", [0, 17])
def test_config5_planes_at_size(golden, k):
    """Two of the 32 planes of BASELINE config 5 at 2048^2 (K = 3 source points): pins the oracle -- pupil with a
    large defocus coefficient, field, accumulation, post-process -- at the size the GPU test compares at."""
    from lithographysimulator_amd.synthetic import bernoulli_mask
    g = golden("g9_config5_stack.npz")
    pn = 2048
    torch.set_num_threads(8)
    mft = O.mask_spectrum(bernoulli_mask(pn), PS, WL)
    ab = list(DEMO_AB); ab[4] = float(g["defocus_nm"][k])
    pf = O.pupil_function(f16(ab), pn, NA, WL)
    eps, N = O.calculate_epsilon_n(4 / pn, PS, WL)
    raw = O.abbe_raw(mft, pf, torch.from_numpy(g["shifts"]), N)
    assert float((crop_center(raw, 64).double() - torch.from_numpy(g["raw_crop"][k]).double()).abs().max()
                 / g["raw_max"][k]) < TOL_IMAGE_MAX
    assert np.allclose(raw.double().sum(1).numpy(), g["raw_rowsum"][k], rtol=1e-5)
    final = O.post_process(raw, eps)
    assert tuple(final.shape) == tuple(g["final_shape"])
    assert float((crop_center(final, 64).double() - torch.from_numpy(g["final_crop"][k]).double()).abs().max()
                 / g["final_max"][k]) < TOL_IMAGE_MAX
    assert abs(float(final.double().sum()) / float(g["final_sum"][k]) - 1) < 1e-5


# ---------------------------------------------------------------- G8 (coarse fp16 grid at 8192)
@pytest.mark.parametrize("pn", [4096, 8192])
def test_pupil_support_at_large_sizes(golden, pn):
    """At pn = 8192 the sigma step 4/8192 is finer than fp16 can hold near 1, so the reference's r <= 1 support
    is 4099 wide instead of pn/2 + 1; the oracle's arange recipe must reproduce that bit for bit."""
    import hashlib
    g = golden("g8_large_pupils.npz")
    W = O.wavefront_error(f16([0, 0, 0, 0, 100]), pn, NA, WL)
    assert np.array_equal(np.frombuffer(hashlib.sha256(W.numpy().tobytes()).digest(), dtype=np.uint8),
                          g[f"Wsha_defocus_p100_{pn}"])
    phi = O.pupil_from_wavefront(W, pn)
    nz = phi != 0
    assert int(nz.sum()) == int(g[f"nz_defocus_p100_{pn}"])
    assert np.array_equal(nz.sum(1).to(torch.int32).numpy(), g[f"rowcount_defocus_p100_{pn}"])
    assert np.array_equal(phi[::64, ::64].numpy(), g[f"phisub_defocus_p100_{pn}"])
