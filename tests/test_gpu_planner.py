"""Boundary and randomized differential tests of the launch PLANNER (csrc/abbe_engine.hip: plan_abbe + the coarse-grid
gates of abbe_accumulate) against the CPU oracle, through the C ABI.

The reference accepts ANY pupil function and ANY shift (imageformation.py:32-45, 62-67): whichever kernel family the
planner picks -- natural-box wave kernels, radix-16 fall-backs, the coarse grid, general (wrapping) mode -- the image must
be the reference's.  The gates that decide are fed here from both sides of every threshold, with the decision asserted
through litho_abbe_last_plan, and a seeded fuzz walks the rest (random supports, shifts, stacks and planner options,
scratch poisoned with NaN, `out` pre-filled).  Options travel through litho_abbe_options (no os.environ)."""
import numpy as np
import pytest
import torch

from helpers import NA, PS, TOL_IMAGE_L2, TOL_IMAGE_MAX, WL, f16, rel_l2, rel_max

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def L():
    import lithographysimulator_amd as L
    from lithographysimulator_amd import _native as nat
    assert nat.lib().litho_target_arch() == b"gfx950"
    return L


def O():
    from oracle import abbe_oracle
    return abbe_oracle


def nat():
    from lithographysimulator_amd import _native
    return _native


def _mask_spectrum(L, dev, pn):
    from lithographysimulator_amd.synthetic import bernoulli_mask
    return L.Mask(bernoulli_mask(pn), PS, dev).fraunhofer(WL, True)


def _disk(L, dev, pn, defocus=60):
    return L.Pupil(pn, WL, NA, f16([0, 0, 0, 0, defocus]), dev).generatePupilFunction().clone()


def _rand_c64(gen, *shape):
    return torch.polar(0.5 + 0.5 * torch.rand(*shape, generator=gen), 6.2831853 * torch.rand(*shape, generator=gen)).to(torch.complex64)


def _strided_points(L, dev, pn, K, sin=0.4, sout=0.8):
    sh = L.sourceShifts(L.LightSource(sin, sout, pn, NA, device=dev).generateAnnular(), pn)
    return sh[(torch.arange(K, device=dev) * sh.shape[0]) // K].contiguous()


def _oracle_chunked(o, mft, pupil, shifts, N, workers=16):
    """oracle.abbe_raw (the reference's op chain, sequential fp32 per chunk) over `workers` contiguous chunks of the list on a
    thread pool (torch's CPU ops release the GIL), partial images added in float64: the same sum to ~1e-7, in a sixteenth
    of the time -- the oracle's FFTs at these sizes run on one core each."""
    from concurrent.futures import ThreadPoolExecutor
    S = shifts.shape[0]
    bounds = [(i * S) // workers for i in range(workers + 1)]
    chunks = [shifts[a:b] for a, b in zip(bounds[:-1], bounds[1:]) if b > a]
    old = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        with ThreadPoolExecutor(max_workers=workers) as pool:
            parts = list(pool.map(lambda sh: o.abbe_raw(mft, pupil, sh, N).double(), chunks))
    finally:
        torch.set_num_threads(old)
    return torch.stack(parts).sum(0)


def _check(got, ref, what):
    e, l2 = rel_max(got, ref), rel_l2(got, ref)
    print(f"{what}: rel-to-max {e:.2e}, rel-L2 {l2:.2e}")
    assert e < TOL_IMAGE_MAX and l2 < TOL_IMAGE_L2, what


# ------------------------------------------------------------------ coarse-grid refusal: box corners
@pytest.mark.parametrize("pn,K", [(512, 8), (1024, 6), (2048, 3)])
def test_square_pupil_with_set_corners_leaves_the_coarse_grid(L, dev, pn, K):
    """A square aperture filling |k| <= pn/4 has the natural box but SET CORNERS: the Nyquist-line correction of the
    coarse grid (products of opposite box edges only) does not cover the corner coefficient, so the planner must stay
    on the direct path -- asked politely (coarse = 1) or firmly (2) -- and the wave kernels must take a full box."""
    o = O()
    N, c, h = 2 * pn, pn // 2, pn // 4
    gen = torch.Generator().manual_seed(pn)
    pupil = torch.zeros(pn, pn, dtype=torch.complex64)
    pupil[c - h:c + h + 1, c - h:c + h + 1] = _rand_c64(gen, 2 * h + 1, 2 * h + 1)
    mft = _mask_spectrum(L, dev, pn)
    sel = _strided_points(L, dev, pn, K)
    ref = o.abbe_raw(mft.cpu(), pupil, sel.cpu(), N)
    for coarse in (1, 2):
        got = L.abbeIntensity(mft, pupil.to(dev), sel, N, options={"coarse": coarse}).cpu()
        plan = nat().last_plan()
        assert plan["coarse_grid"] == 0 and plan["natural_box"] == 1 and plan["general"] == 0, plan
        assert (plan["box_rows"], plan["box_cols"]) == (2 * h + 1, 2 * h + 1), plan
        _check(got, ref, f"{pn}^2 square pupil, coarse={coarse}, y-pass {nat().last_kernels()[1]}")
    # one corner alone is enough to refuse; none, and the same call runs on the coarse grid
    for corner, expect in (((c + h, c - h), 0), (None, 1)):
        p2 = _disk(L, dev, pn)
        if corner:
            p2[corner] = 0.7 - 0.2j
        got = L.abbeIntensity(mft, p2, sel, N, options={"coarse": 2}).cpu()
        assert nat().last_plan()["coarse_grid"] == expect, (corner, nat().last_plan())
        _check(got, o.abbe_raw(mft.cpu(), p2.cpu(), sel.cpu(), N), f"{pn}^2 disk, corner {corner}")


# ------------------------------------------------------------------ coarse-grid refusal: box-edge support 128 / 129
@pytest.mark.parametrize("pn,K", [(1024, 6), (2048, 3)])
@pytest.mark.parametrize("edge", ["columns", "rows"])
@pytest.mark.parametrize("length", [128, 129])
def test_box_edge_support_at_the_coarse_grid_limit(L, dev, pn, K, edge, length):
    """Disk + a rim on the two opposite edges of the natural box whose joint support is exactly 128 samples (the most
    k_nyquist_edges handles: coarse grid, with 255 non-zero Nyquist-line coefficients) or 129 (must leave it)."""
    o = O()
    N, c, h = 2 * pn, pn // 2, pn // 4
    gen = torch.Generator().manual_seed(1000 * pn + length)
    pupil = _disk(L, dev, pn).cpu()
    lo = c - 64
    a = _rand_c64(gen, length)              # edge +h: the whole support
    b = _rand_c64(gen, 51)                  # edge -h: an inner part of it
    if edge == "columns":
        pupil[lo:lo + length, c + h] = a
        pupil[c - 30:c + 21, c - h] = b
    else:
        pupil[c + h, lo:lo + length] = a
        pupil[c - h, c - 30:c + 21] = b
    mft = _mask_spectrum(L, dev, pn)
    sel = _strided_points(L, dev, pn, K)
    ref = o.abbe_raw(mft.cpu(), pupil, sel.cpu(), N)
    for coarse in (2, 0):
        got = L.abbeIntensity(mft, pupil.to(dev), sel, N, options={"coarse": coarse}).cpu()
        plan = nat().last_plan()
        assert plan["natural_box"] == 1 and plan["general"] == 0, plan
        assert plan["coarse_grid"] == (1 if coarse == 2 and length <= 128 else 0), (length, coarse, plan)
        _check(got, ref, f"{pn}^2 rim on {edge}, support {length}, coarse={coarse} -> coarse_grid {plan['coarse_grid']}")


def test_stack_in_which_one_plane_alone_violates(L, dev):
    """The plan words are taken over ALL planes of a stack: if plane 1 alone has a 129-sample edge (or a set corner), the
    whole stack leaves the coarse grid; with a 128-sample edge on plane 1 only it stays, and the edge sums of planes 0
    and 2 (one sample per edge) use the joint support."""
    o = O()
    pn, K = 1024, 4
    N, c, h = 2 * pn, pn // 2, pn // 4
    gen = torch.Generator().manual_seed(77)
    mft = _mask_spectrum(L, dev, pn)
    sel = _strided_points(L, dev, pn, K)
    for what, expect in (("edge128", 1), ("edge129", 0), ("corner", 0)):
        stack = torch.stack([_disk(L, dev, pn, d).cpu() for d in (-80, 20, 120)])
        if what == "corner":
            stack[1, c - h, c - h] = 0.3 + 0.4j
        else:
            n = 128 if what == "edge128" else 129
            stack[1, c - 64:c - 64 + n, c - h] = _rand_c64(gen, n)
            stack[1, c - 10:c + 10, c + h] = _rand_c64(gen, 20)
        for pc in (1, 2):
            got = L.abbeIntensity(mft, stack.to(dev), sel, N, options={"coarse": 2, "plane_chunk": pc}).cpu()
            plan = nat().last_plan()
            assert plan["coarse_grid"] == expect and plan["natural_box"] == 1, (what, plan)
            for k in range(3):
                _check(got[k], o.abbe_raw(mft.cpu(), stack[k], sel.cpu(), N), f"stack {what}, plane chunk {pc}, plane {k}")


# ------------------------------------------------------------------ natural box: pn/4 and pn/4 + 1 on each side
@pytest.mark.parametrize("pn", [512, 1024])
@pytest.mark.parametrize("side", ["col_hi", "col_lo", "row_hi", "row_lo"])
def test_support_box_one_sample_inside_and_outside_the_natural_box(L, dev, pn, side):
    """One extra pupil sample AT |k| = pn/4 (still the natural box: wave kernels, coarse grid) and at pn/4 + 1 (the box
    check must send the call to the radix-16 kernels on the direct path) on each of the four sides."""
    o = O()
    N, c, h = 2 * pn, pn // 2, pn // 4
    mft = _mask_spectrum(L, dev, pn)
    sel = _strided_points(L, dev, pn, 6)
    for beyond in (0, 1):
        pupil = _disk(L, dev, pn).cpu()
        k = h + beyond
        pos = {"col_hi": (c + 5, c + k), "col_lo": (c - 7, c - k), "row_hi": (c + k, c + 9), "row_lo": (c - k, c - 3)}[side]
        pupil[pos] = -0.6 + 0.5j
        ref = o.abbe_raw(mft.cpu(), pupil, sel.cpu(), N)
        for coarse in (2, 0):
            got = L.abbeIntensity(mft, pupil.to(dev), sel, N, options={"coarse": coarse}).cpu()
            plan = nat().last_plan()
            assert plan["natural_box"] == 1 - beyond and plan["general"] == 0, (side, beyond, plan)
            assert plan["coarse_grid"] == (1 if coarse == 2 and not beyond else 0), (side, beyond, plan)
            assert plan["wave_ypass"] == 1 - beyond, (side, beyond, plan)
            assert max(plan["box_rows"], plan["box_cols"]) == 2 * h + 1 + beyond, plan
            _check(got, ref, f"{pn}^2 extra sample {side} at pn/4+{beyond}, coarse={coarse}, kernels {nat().last_kernels()}")


# ------------------------------------------------------------------ source-count threshold of the default mode
@pytest.mark.parametrize("pn,s_min", [(512, 512), (1024, 256), (2048, 96)])
def test_source_count_threshold_of_the_coarse_grid(L, dev, pn, s_min):
    """Default mode (coarse = 1): S = s_min - 1 stays on the direct path, S = s_min takes the coarse grid; both are the
    reference's image (the oracle's op chain over the same consecutive points)."""
    o = O()
    N = 2 * pn
    mft = _mask_spectrum(L, dev, pn)
    pupil = _disk(L, dev, pn, 100)
    sh = L.sourceShifts(L.LightSource(0.4, 0.8, pn, NA, device=dev).generateAnnular(), pn)
    lo = sh.shape[0] // 3
    sel = sh[lo:lo + s_min].contiguous()
    ref_short = _oracle_chunked(o, mft.cpu(), pupil.cpu(), sel[:-1].cpu(), N)
    ref_full = ref_short + o.abbe_raw(mft.cpu(), pupil.cpu(), sel[-1:].cpu(), N).double()
    for S, ref, expect in ((s_min - 1, ref_short, 0), (s_min, ref_full, 1)):
        got = L.abbeIntensity(mft, pupil, sel[:S].contiguous(), N, options={"coarse": 1}).cpu()
        plan = nat().last_plan()
        assert plan["coarse_grid"] == expect and plan["natural_box"] == 1, (S, plan)
        _check(got, ref, f"{pn}^2 S = {S} (threshold {s_min}) -> coarse_grid {expect}")


# ------------------------------------------------------------------ wrapping: one sample short of it and one past
@pytest.mark.parametrize("pn", [256, 512])
@pytest.mark.parametrize("axis,sign", [(0, -1), (0, 1), (1, -1), (1, 1)])
def test_shift_one_sample_short_of_wrapping_and_one_past(L, dev, pn, axis, sign):
    """The disk's box is rows/columns [c - h, c + h]: a shift of -(c - h) puts its first row on row 0 (no wrap: pruned
    box mode, coarse grid allowed, mask samples read at the very edge of the grid), one more wraps (general mode: roll
    kept on P, modular gather).  Same on the high side (last row on row pn - 1) and along x."""
    o = O()
    N, c, h = 2 * pn, pn // 2, pn // 4
    mft = _mask_spectrum(L, dev, pn)
    pupil = _disk(L, dev, pn)
    edge = -(c - h) if sign < 0 else (pn - 1) - (c + h)           # the last shift that does not wrap
    for past in (0, 1):
        big = edge + sign * past
        pts = [[3, -5], [0, 0], [-11, 17]]
        pts[1][axis] = big
        pts.append([big if axis == 0 else 2, big if axis == 1 else -4])
        sel = torch.tensor(pts, dtype=torch.int32, device=dev)
        ref = o.abbe_raw(mft.cpu(), pupil.cpu(), sel.cpu(), N)
        for coarse in (2, 0):
            got = L.abbeIntensity(mft, pupil, sel, N, options={"coarse": coarse}).cpu()
            plan = nat().last_plan()
            assert plan["general"] == past, (axis, sign, past, plan)
            assert plan["coarse_grid"] == (1 if coarse == 2 and not past else 0), (axis, sign, past, plan)
            _check(got, ref, f"{pn}^2 shift {big} on axis {axis} (wraps: {past}), coarse={coarse}")


# ------------------------------------------------------------------ mask sizes that are neither N nor N / 2: embedded evaluation
@pytest.mark.parametrize("pn,ps,pe", [(200, 25, 256), (1000, 25, 1024), (1500, 25, 2048), (768, 25, 1024), (300, 10, 512), (502, 25, 512), (2000, 25, 2048),
                                      (3000, 25, 4096)])
def test_odd_mask_sizes_run_embedded_and_match_the_oracle(L, dev, monkeypatch, pn, ps, pe):
    """A 1000^2 (1500^2, 768^2 ...) mask is neither N nor N / 2: instead of the generic kernels the library pads mask
    spectrum and pupil into the next such grid inside its workspace, runs the power-of-two kernels (coarse grid included)
    and adds the centre of the padded image to the caller's.  Same sum term by term: single plane, a stack, a pre-filled `out`, through abbeImage, with
    a PlanCache -- against the oracle and against the un-embedded generic evaluation."""
    o = O()
    from lithographysimulator_amd.synthetic import bernoulli_mask
    gen = torch.Generator().manual_seed(pn)
    geo = (torch.rand(pn, pn, generator=gen) < 0.5).to(torch.int16)
    mask = L.Mask(geo, ps, dev)
    mft = mask.fraunhofer(WL, True)
    eps, N = mask.calculateEpsilonN(mask.deltaK, ps, WL)
    assert L.embeddedSize(pn, N) == pe
    pupil = L.Pupil(pn, WL, NA, f16([0, 0, 0.01, 0, 80, 0.01]), dev).generatePupilFunction()
    bitmap = L.LightSource(0.4, 0.8, pn, NA, device=dev).generateAnnular()
    sh = L.sourceShifts(bitmap, pn)
    K = 12 if pn <= 1000 else 5 if pn <= 1500 else 3
    sel = sh[(torch.arange(K, device=dev) * sh.shape[0]) // K].contiguous()
    ref = o.abbe_raw(mft.cpu(), pupil.cpu(), sel.cpu(), N)
    for coarse in (1, 2, 0):
        got = L.abbeIntensity(mft, pupil, sel, N, options={"coarse": coarse}).cpu()
        plan = nat().last_plan()
        assert plan["general"] == 0 and plan["variant"] >= 0, plan               # a specialised kernel family, at the padded size
        assert plan["coarse_grid"] == (1 if coarse == 2 and 2 * pe == N else 0), (coarse, plan)
        _check(got, ref, f"{pn}^2 (N {N}) embedded in {pe}^2, coarse={coarse}, kernels {nat().last_kernels()}")
    plain = L.abbeIntensity(mft, pupil, sel, N, options={"embed": 0}).cpu()
    assert nat().last_plan()["variant"] == (-1 if pn & (pn - 1) else nat().last_plan()["variant"])
    _check(plain, ref, f"{pn}^2 un-embedded")
    # a stack, accumulated into a pre-filled buffer
    stack = L.throughFocusPupils(pn, WL, NA, f16([0, 0, 0.01, 0, 80, 0.01]), [-60.0, 40.0], dev)
    pre = torch.rand(2, pn, pn, generator=gen).to(dev) * float(ref.max()) * 0.2
    both = L.abbeIntensity(mft, stack, sel, N, out=pre.clone()).cpu()
    for k in range(2):
        want = pre[k].cpu().double() + o.abbe_raw(mft.cpu(), stack[k].cpu(), sel.cpu(), N).double()
        _check(both[k], want, f"{pn}^2 stack plane {k}, pre-filled out")
    if pn == 200:
        # more planes than the embedding pads at once (four): chunk boundaries of the padded stack
        st6 = L.throughFocusPupils(pn, WL, NA, f16([0, 0, 0.01, 0, 80, 0.01]), [-100.0, -60.0, -20.0, 20.0, 60.0, 100.0], dev)
        six = L.abbeIntensity(mft, st6, sel, N, options={"coarse": 2}).cpu()
        for k in range(6):
            _check(six[k], o.abbe_raw(mft.cpu(), st6[k].cpu(), sel.cpu(), N), f"{pn}^2 six-plane stack, plane {k}")
    # end to end, with a PlanCache (second call plans from the record, on the padded grid)
    few = torch.zeros_like(bitmap)
    pts = torch.argwhere(bitmap)[(torch.arange(K, device=dev) * sh.shape[0]) // K]
    few[pts[:, 0], pts[:, 1]] = 1
    cache = L.PlanCache()
    img1 = L.abbeImage(mask, mft, pupil, few, ps, mask.deltaK, WL, True, dev, plan_cache=cache)
    img2 = L.abbeImage(mask, mft, pupil, few, ps, mask.deltaK, WL, True, dev, plan_cache=cache)
    assert nat().last_plan()["planned_from_record"] == 1 and cache.record.pn == pn and torch.equal(img1, img2)
    want = o.post_process(o.abbe_raw(mft.cpu(), pupil.cpu(), o.source_shifts(few.cpu(), pn), N), eps)
    assert img1.shape == want.shape
    _check(img1.cpu(), want, f"{pn}^2 abbeImage end to end")
    # a record made for the embedded run is not reused with the embedding off (its edge words were looked up for the padded
    # grid), and the other way round
    img_off = L.abbeImage(mask, mft, pupil, few, ps, mask.deltaK, WL, True, dev, plan_cache=cache, options={"embed": 0})
    assert nat().last_plan()["planned_from_record"] == 0
    _check(img_off.cpu(), want, f"{pn}^2 abbeImage, same PlanCache, embedding off")
    L.abbeImage(mask, mft, pupil, few, ps, mask.deltaK, WL, True, dev, plan_cache=cache)
    assert nat().last_plan()["planned_from_record"] == 0
    img3 = L.abbeImage(mask, mft, pupil, few, ps, mask.deltaK, WL, True, dev)            # the default call: device-side source count
    assert nat().last_plan()["planned_from_record"] == 0 and torch.equal(img3, img1)
    norm = L.abbeImage(mask, mft, pupil, few, ps, mask.deltaK, WL, True, dev, normalize=True)
    assert rel_max((norm * K).cpu(), img1.cpu()) < 1e-6


@pytest.mark.parametrize("embed", [1, 0])
def test_odd_mask_sizes_vs_the_reference_itself(golden, L, dev, embed):
    """Golden g16: 200^2, 1000^2, 1500^2 masks through the REFERENCE's own abbeImage (its torch.fft path takes any size and
    knows nothing of padded grids) -- strided points at each size and 600 consecutive points at 1000^2 (long enough for the
    coarse grid of the padded 1024^2).  The engine's embedded evaluation (and, embed = 0, its generic kernels) against raw and
    final images: full image at 200^2 (final 198 x 198, the reference's pad arithmetic), crops and every row / column sum above."""
    from lithographysimulator_amd.synthetic import bernoulli_mask
    g = golden("g16_odd_sizes.npz")
    for tag, pn in (("p200", 200), ("p1000", 1000), ("p1500", 1500), ("run1000", 1000)):
        mask = L.Mask(bernoulli_mask(pn), PS, dev)
        mft = mask.fraunhofer(WL, True)
        eps, N = mask.calculateEpsilonN(mask.deltaK, PS, WL)
        pf = L.Pupil(pn, WL, NA, f16([0, 0, 0.01, 0, 100, 0.01, 0, 0.01, 0.01, 0.01]), dev).generatePupilFunction()
        sh = L.sourceShifts(L.LightSource(0.4, 0.8, pn, NA, device=dev).generateAnnular(), pn)
        if tag == "run1000":
            lo, hi, S = (int(v) for v in g["run1000_range"])
            assert sh.shape[0] == S
            sel = sh[lo:hi]
            assert np.array_equal(sel[[0, -1]].cpu().numpy(), g["run1000_first_last_shift"])
        else:
            assert N == int(g[f"{tag}_N"]) and sh.shape[0] == int(g[f"{tag}_S_full"])
            sel = torch.from_numpy(g[f"{tag}_shifts"]).to(dev)
        raw = L.abbeIntensity(mft, pf, sel, N, options={"embed": embed})
        plan = nat().last_plan()
        assert plan["general"] == 0 and (plan["variant"] >= 0) == bool(embed), (tag, plan)
        if tag == "run1000":
            assert plan["coarse_grid"] == embed, plan                         # 600 points >= 256 at the padded 1024^2
        final = L.postProcess(raw, eps).cpu()
        raw = raw.cpu()
        assert tuple(final.shape) == tuple(g[f"{tag}_final_shape"])
        for kind, img in (("raw", raw), ("final", final)):
            mx = float(g[f"{tag}_{kind}_max"])
            crop = min(128, img.shape[0])
            c0 = img.shape[0] // 2 - crop // 2
            e = float((img[c0:c0 + crop, c0:c0 + crop].double() - torch.from_numpy(g[f"{tag}_{kind}_crop"]).double()).abs().max() / mx)
            if f"{tag}_{kind}_image" in g.files:
                e = max(e, float((img.double() - torch.from_numpy(g[f"{tag}_{kind}_image"]).double()).abs().max() / mx))
            if f"{tag}_{kind}_stride8" in g.files:
                e = max(e, float(np.abs(img[::8, ::8].numpy().astype(np.float64) - g[f"{tag}_{kind}_stride8"]).max() / mx))
            print(f"{tag} ({pn}^2, embed {embed}) vs the reference, {kind}: {e:.2e} of the maximum")
            assert e < TOL_IMAGE_MAX, (tag, kind, e)
            assert np.allclose(img.double().sum(1).numpy(), g[f"{tag}_{kind}_rowsum"], rtol=2e-5, atol=2e-6 * float(np.abs(g[f"{tag}_{kind}_rowsum"]).max()))
            assert np.allclose(img.double().sum(0).numpy(), g[f"{tag}_{kind}_colsum"], rtol=2e-5, atol=2e-6 * float(np.abs(g[f"{tag}_{kind}_colsum"]).max()))
            assert abs(float(img.double().sum()) / float(g[f"{tag}_{kind}_sum"]) - 1) < 2e-6


@pytest.mark.parametrize("pn", [200, 1000])
def test_embedded_evaluation_refuses_shifts_that_wrap_the_original_grid(L, dev, pn):
    """The reference rolls the pupil modulo ITS grid (imageformation.py:63).  In the padded grid a large shift has room
    that the original grid does not: the engine tests the wrap on the ORIGINAL size and runs such a list on the general path
    at the caller's size -- one sample short of wrapping runs embedded."""
    o = O()
    gen = torch.Generator().manual_seed(3 * pn)
    mask = L.Mask((torch.rand(pn, pn, generator=gen) < 0.5).to(torch.int16), PS, dev)
    mft = mask.fraunhofer(WL, True)
    eps, N = mask.calculateEpsilonN(mask.deltaK, PS, WL)
    pupil = L.Pupil(pn, WL, NA, f16([0, 0, 0, 0, 50]), dev).generatePupilFunction()
    c, h = pn // 2, pn // 4
    edge = -(c - h)                                                     # the pupil's first row lands on row 0
    for past, expect_general in ((0, 0), (1, 1)):
        sel = torch.tensor([[3, -5], [edge - past, 7], [-11, 17]], dtype=torch.int32, device=dev)
        pre = torch.full((pn, pn), 2.5, device=dev)
        got = L.abbeIntensity(mft, pupil, sel, N, out=pre.clone(), options={"coarse": 2}).cpu()
        plan = nat().last_plan()
        assert plan["general"] == expect_general, (past, plan)
        assert plan["box_rows"] == (pn if past else 2 * h + 1), plan          # general mode runs at the caller's size
        want = 2.5 + o.abbe_raw(mft.cpu(), pupil.cpu(), sel.cpu(), N).double()
        _check(got, want, f"{pn}^2 shift {edge - past} (wraps the original grid: {past})")


# ------------------------------------------------------------------ shifted (off-axis) sources: the list is split
@pytest.mark.parametrize("pn,K", [(512, 420), (1000, 300), (256, 40)])
def test_shifted_source_is_split_into_a_fast_and_a_general_part(L, dev, pn, K):
    """LightSource(shiftX, shiftY) (lightsource.py:5) moves the whole source: for part of its points the rolled pupil wraps
    around the grid, which only the general path reproduces (4x the time per point).  The list is split on the device -- stable,
    deterministic -- so that only the wrapping points pay: against the oracle (whose torch.roll wraps), against the unsplit
    general evaluation, at a power-of-two size, at an embedded one (the non-wrapping part runs padded, the other at the caller's
    size) and below the automatic threshold with split = 2; all-wrapping and none-wrapping lists as the degenerate cases."""
    o = O()
    from lithographysimulator_amd.synthetic import bernoulli_mask
    mask = L.Mask(bernoulli_mask(pn), PS, dev)
    mft = mask.fraunhofer(WL, True)
    eps, N = mask.calculateEpsilonN(mask.deltaK, PS, WL)
    pupil = L.Pupil(pn, WL, NA, f16([0, 0, 0.01, 0, 70]), dev).generatePupilFunction()
    sh = L.sourceShifts(L.LightSource(0.4, 0.8, pn, NA, shiftX=0.25, shiftY=-0.5, device=dev).generateAnnular(), pn)
    sel = sh[(torch.arange(K, device=dev) * sh.shape[0]) // K].contiguous()
    c, h = pn // 2, pn // 4
    wraps = (sel[:, 0] < -(c - h)) | (sel[:, 0] > (pn - 1) - (c + h)) | (sel[:, 1] < -(c - h)) | (sel[:, 1] > (pn - 1) - (c + h))
    nw = int(wraps.sum())
    assert 0 < nw < K, (nw, K)
    ref = _oracle_chunked(o, mft.cpu(), pupil.cpu(), sel.cpu(), N)
    mode = 2 if K < 256 else 1
    got = L.abbeIntensity(mft, pupil, sel, N, options={"split": mode}).cpu()
    plan = nat().last_plan()
    assert plan["planned_from_record"] == 2 and plan["general"] == 1, plan                  # split; the wrapping part ran last
    _check(got, ref, f"{pn}^2 shifted source, {nw} of {K} points wrap, split")
    whole = L.abbeIntensity(mft, pupil, sel, N, options={"split": 0}).cpu()
    plan0 = nat().last_plan()
    assert plan0["planned_from_record"] == 0 and plan0["general"] == 1, (plan0, plan)
    _check(whole, ref, f"{pn}^2 shifted source, unsplit general evaluation")
    # a stack, into a pre-filled buffer
    stack = L.throughFocusPupils(pn, WL, NA, f16([0, 0, 0.01, 0, 70]), [-50.0, 30.0], dev)
    pre = torch.full((2, pn, pn), 0.1 * float(ref.max()), device=dev)
    both = L.abbeIntensity(mft, stack, sel, N, out=pre.clone(), options={"split": mode}).cpu()
    assert nat().last_plan()["planned_from_record"] == 2
    for k in range(2):
        _check(both[k], pre[k].cpu().double() + _oracle_chunked(o, mft.cpu(), stack[k].cpu(), sel.cpu(), N), f"{pn}^2 shifted source, stack plane {k}")
    # degenerate lists: every point wraps / none does (no split happens for the second: nothing wraps)
    only_w, only_n = sel[wraps].contiguous(), sel[~wraps].contiguous()
    a = L.abbeIntensity(mft, pupil, only_w, N, options={"split": 2}).cpu()
    assert nat().last_plan()["planned_from_record"] == 2 and nat().last_plan()["general"] == 1
    b = L.abbeIntensity(mft, pupil, only_n, N, options={"split": 2}).cpu()
    assert nat().last_plan()["planned_from_record"] == 0 and nat().last_plan()["general"] == 0
    _check(a.double() + b.double(), ref, f"{pn}^2 shifted source, wrapping and non-wrapping halves separately")
    # end to end through abbeImage with the shifted bitmap
    few = torch.zeros((pn, pn), dtype=torch.int64, device=dev)
    few[(sel[:, 0] + c).long(), (sel[:, 1] + c).long()] = 1
    img = L.abbeImage(mask, mft, pupil, few, PS, mask.deltaK, WL, True, dev, options={"split": mode}).cpu()
    _check(img, o.post_process(ref.float(), eps), f"{pn}^2 shifted source through abbeImage")


def test_planned_calls_keep_the_split_of_a_shifted_source(L, dev):
    """Round-4 advice: a PlanCache used to make shifted-source sequences SLOWER than uncached calls -- the planning call split
    the list, every later call found the whole-list extents in the record and ran all points on the general path.  The record now
    carries the split's outcome; a planned call re-runs the three split kernels (same list, same result, no read-back) and plans
    both parts from it: last_plan reports 3 (split, from the record), the image equals the planning call's bit for bit, with
    another mask too, and equals the oracle."""
    o = O()
    from lithographysimulator_amd.synthetic import bernoulli_mask, lines_mask
    pn, K = 512, 420
    mask = L.Mask(bernoulli_mask(pn), PS, dev)
    mft = mask.fraunhofer(WL, True)
    eps, N = mask.calculateEpsilonN(mask.deltaK, PS, WL)
    pupil = L.Pupil(pn, WL, NA, f16([0, 0, 0.01, 0, 70]), dev).generatePupilFunction()
    sh = L.sourceShifts(L.LightSource(0.4, 0.8, pn, NA, shiftX=0.25, shiftY=-0.5, device=dev).generateAnnular(), pn)
    sel = sh[(torch.arange(K, device=dev) * sh.shape[0]) // K].contiguous()
    cache = L.PlanCache()
    first, S = L.abbeIntensity(mft, pupil, sel, N, plan=cache)
    assert S == K and nat().last_plan()["planned_from_record"] == 2, nat().last_plan()          # planned afresh, list split
    again, _ = L.abbeIntensity(mft, pupil, sel, N, plan=cache)
    plan = nat().last_plan()
    assert plan["planned_from_record"] == 3 and plan["general"] == 1, plan                       # split again, from the record
    assert torch.equal(first, again)
    _check(again.cpu(), _oracle_chunked(o, mft.cpu(), pupil.cpu(), sel.cpu(), N), "planned call of a shifted source")
    mft2 = L.Mask(lines_mask(pn), PS, dev).fraunhofer(WL, True)
    other, _ = L.abbeIntensity(mft2, pupil, sel, N, plan=cache)
    assert nat().last_plan()["planned_from_record"] == 3
    _check(other.cpu(), _oracle_chunked(o, mft2.cpu(), pupil.cpu(), sel.cpu(), N), "planned call of a shifted source, another mask")
    # the launches of a planned split: as many as the planning call made
    unplanned = L.abbeIntensity(mft2, pupil, sel, N)
    assert nat().last_plan()["planned_from_record"] == 2 and torch.equal(unplanned, other)
    # a record this library version did not write (valid flag and sizes right, words without the format tag -- e.g. kept across an
    # upgrade) is ignored: the call plans afresh instead of unpacking garbage
    forged = L.PlanCache()
    forged.record.valid, forged.record.pn, forged.record.N, forged.record.planes = 1, pn, N, 1
    for i, v in enumerate([128, 384, 128, 384, -10, 10, -10, 10, K, 0, 0, 0, 0, 0, pn, 0]):      # the round-4 layout of the 16 words
        forged.record.words[i] = v
    f, _ = L.abbeIntensity(mft, pupil, sel, N, plan=forged)
    assert nat().last_plan()["planned_from_record"] == 2 and torch.equal(f, first)
    # a record made WITHOUT the split (options) stays unsplit when reused: general path for every point, same image to rounding
    cache0 = L.PlanCache()
    a0, _ = L.abbeIntensity(mft, pupil, sel, N, plan=cache0, options={"split": 0})
    b0, _ = L.abbeIntensity(mft, pupil, sel, N, plan=cache0)
    assert nat().last_plan()["planned_from_record"] == 1 and nat().last_plan()["general"] == 1
    assert rel_max(b0.cpu(), first.cpu()) < 2e-6 and torch.equal(a0, b0)


@pytest.mark.parametrize("pn,K", [(2048, 12), (3000, 10)])
def test_shifted_source_split_at_the_large_sizes(L, dev, pn, K):
    """The split at 2048^2 and at an embedded 3000^2 (padded grid 4096; round-4 review, weak 1c): there a general-mode T item is 4x
    a pruned one and the T regions of BOTH carves are cut short by the two lists.  Against the oracle; the dry run of the same
    call (litho_abbe_plan_dry_run) asserts the regions apart."""
    o = O()
    from lithographysimulator_amd.synthetic import bernoulli_mask
    mask = L.Mask(bernoulli_mask(pn), PS, dev)
    mft = mask.fraunhofer(WL, True)
    eps, N = mask.calculateEpsilonN(mask.deltaK, PS, WL)
    pupil = L.Pupil(pn, WL, NA, f16([0, 0, 0.01, 0, 70]), dev).generatePupilFunction()
    sh = L.sourceShifts(L.LightSource(0.4, 0.8, pn, NA, shiftX=0.25, shiftY=-0.5, device=dev).generateAnnular(), pn)
    sel = sh[(torch.arange(K, device=dev) * sh.shape[0]) // K].contiguous()
    c, h = pn // 2, pn // 4
    nzr = torch.nonzero(pupil.abs().sum(1) > 0).flatten()
    nzc = torch.nonzero(pupil.abs().sum(0) > 0).flatten()
    r0, r1, c0, c1 = int(nzr[0]), int(nzr[-1]), int(nzc[0]), int(nzc[-1])
    wraps = (sel[:, 0] + r0 < 0) | (sel[:, 0] + r1 > pn - 1) | (sel[:, 1] + c0 < 0) | (sel[:, 1] + c1 > pn - 1)
    nw = int(wraps.sum())
    assert 0 < nw < K, (nw, K)
    ref = _oracle_chunked(o, mft.cpu(), pupil.cpu(), sel.cpu(), N, workers=K)
    got = L.abbeIntensity(mft, pupil, sel, N, options={"split": 2, "poison": 1}).cpu()
    plan = nat().last_plan()
    assert plan["planned_from_record"] == 2 and plan["general"] == 1, plan
    _check(got, ref, f"{pn}^2 shifted source, {nw} of {K} points wrap, split")
    # the same call through the dry run: split, part 0 on the fast path (embedded for 3000^2), regions apart
    a, b = sel[~wraps], sel[wraps]
    ext = lambda t: [int(t[:, 0].min()), int(t[:, 0].max()), int(t[:, 1].min()), int(t[:, 1].max())]          # noqa: E731
    words = [r0, r1, c0, c1] + ext(sel) + [K, 2**31 - 1, -2**31, 2**31 - 1, -2**31, 0]
    dry = nat().plan_dry_run(pn, N, 1, words, [K - nw, nw] + ext(a) + ext(b), options={"split": 2})
    assert dry.status == 0 and dry.split == 1 and dry.part[0].general == 0 and dry.part[1].general == 1
    assert dry.part[0].run_size == L.embeddedSize(pn, N) and dry.part[1].run_size == pn
    assert dry.part[0].T_region.end <= dry.list_a.offset and dry.part[1].T_region.end <= dry.list_a.offset
    assert dry.part[1].batch == plan["batch"]                     # the part that ran last on the device is the wrapping one


def _plan_words_of(pupils, shifts, pe):
    """The 14 plan words the engine reads back (k_plan_gather + k_plan_finish), computed on the host from the tensors: support box
    of the non-zero pupil samples over all planes, shift extents, count, supports on the natural-box edges of the grid the engine
    will run at (pe), corner flag."""
    INT_MAX, INT_MIN = 2**31 - 1, -2**31
    pn = pupils.shape[-1]
    nz = (pupils.reshape(-1, pn, pn) != 0).any(0)
    rows, cols = torch.nonzero(nz.any(1)).flatten(), torch.nonzero(nz.any(0)).flatten()
    if rows.numel() == 0:
        return None
    w = [int(rows[0]), int(rows[-1]), int(cols[0]), int(cols[-1]), int(shifts[:, 0].min()), int(shifts[:, 0].max()),
         int(shifts[:, 1].min()), int(shifts[:, 1].max()), int(shifts.shape[0])]
    lo, hi = pn // 2 - pe // 4, pn // 2 + pe // 4
    ecols = [c for c in (lo, hi) if 0 <= c < pn]
    er = torch.nonzero(nz[:, ecols].any(1)).flatten() if ecols else torch.zeros(0)
    ec = torch.nonzero(nz[ecols, :].any(0)).flatten() if ecols else torch.zeros(0)
    w += [int(er[0]), int(er[-1])] if er.numel() else [INT_MAX, INT_MIN]
    w += [int(ec[0]), int(ec[-1])] if ec.numel() else [INT_MAX, INT_MIN]
    w.append(int(any(bool(nz[r, c]) for r in ecols for c in ecols)))
    return w


def test_dry_run_predicts_the_plan_the_device_call_reports(L, dev):
    """litho_abbe_plan_dry_run and the real call share their planning code (csrc/abbe_plan.hpp); this ties the CPU sweep of
    tests/test_planner_cpu.py to the engine: for 120 cases of the seeded fuzz (random pupils, shifts, planes and options -- unsplit
    lists) the dry run, fed with plan words computed on the HOST from the tensors, reports exactly the plan that
    litho_abbe_last_plan returns after the device call: path, box, batch, groups, chunk, planes in flight, x-pass kind."""
    checked = 0
    for seed in range(120):
        cs = _fuzz_case(seed)
        if cs["kind"] == "empty":
            continue
        pn = cs["pn"]
        _, N = L.Mask.calculateEpsilonN(None, 4 / pn, cs["ps"], WL)
        opts = dict(cs["opts"], split=0, poison=0)
        stacked = cs["planes"] > 1
        pup = (cs["pupils"] if stacked else cs["pupils"][0]).to(dev)
        L.abbeIntensity(cs["mft"].to(dev), pup, cs["shifts"].to(dev), N, options=opts)
        plan = nat().last_plan()
        pe = L.embeddedSize(pn, N) if opts.get("embed", 1) else pn
        words = _plan_words_of(cs["pupils"], cs["shifts"], pe)
        dry = nat().plan_dry_run(pn, N, cs["planes"], words, None, options=opts, cus=torch.cuda.get_device_properties(dev).multi_processor_count)
        assert dry.status == 0 and dry.split == 0, seed
        p = dry.part[0]
        if p.run_size != pn and cs["planes"] > 4:
            continue        # an embedded stack is padded four planes at a time: last_plan describes the LAST chunk (one plane here)
        tag = f"seed {seed}: pn {pn} N {N} {cs['kind']} planes {cs['planes']} {cs['mode']} {opts} -> device {plan} / dry batch {p.batch} G {p.groups} xchunk {p.xchunk}"
        assert p.present and (p.general, p.coarse, p.wave_y, p.natural_box) == (plan["general"], plan["coarse_grid"], plan["wave_ypass"], plan["natural_box"]), tag
        assert (p.batch, p.groups, p.xchunk, p.planes_in_flight, p.variant, p.xkind) == (
            plan["batch"], plan["groups_per_plane"], plan["xchunk"], plan["planes_in_flight"], plan["variant"], plan["fused_xpass"]), tag
        checked += 1
    assert checked >= 100


def test_environment_variables_remain_a_fallback_and_options_win(L, dev, monkeypatch):
    """Options passed per call beat the LITHO_ABBE_* variables; a field the caller leaves unset falls back to the variable,
    then to the default."""
    pn, N = 512, 1024
    mft = _mask_spectrum(L, dev, pn)
    pupil = _disk(L, dev, pn)
    sel = _strided_points(L, dev, pn, 600)
    base = L.abbeIntensity(mft, pupil, sel, N)
    assert nat().last_plan()["coarse_grid"] == 1                       # default rule: 600 points >= 512
    monkeypatch.setenv("LITHO_ABBE_COARSE", "0")
    monkeypatch.setenv("LITHO_ABBE_BATCH", "7")
    env_only = L.abbeIntensity(mft, pupil, sel, N)
    plan = nat().last_plan()
    assert plan["coarse_grid"] == 0 and plan["batch"] == 7, plan
    both = L.abbeIntensity(mft, pupil, sel, N, options={"coarse": 2})
    plan = nat().last_plan()
    assert plan["coarse_grid"] == 1 and plan["batch"] == 7, plan       # `coarse` from the call, `batch` still from the variable
    with nat().engineOptions(batch=9):
        L.abbeIntensity(mft, pupil, sel, N)
        plan = nat().last_plan()
        assert plan["coarse_grid"] == 0 and plan["batch"] == 9, plan
    assert rel_max(env_only.cpu(), base.cpu()) < 2e-6 and rel_max(both.cpu(), base.cpu()) < 2e-6


# ------------------------------------------------------------------ seeded fuzz
FUZZ_CASES = 320


def _fuzz_pupil(gen, kind, pn):
    c, h = pn // 2, pn // 4
    p = torch.zeros(pn, pn, dtype=torch.complex64)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))          # noqa: E731  inclusive
    if kind == "empty":
        return p
    if kind == "single":
        p[ri(0, pn - 1), ri(0, pn - 1)] = complex(_rand_c64(gen, 1)[0])
        return p
    if kind == "full":
        return _rand_c64(gen, pn, pn)
    if kind == "box":                                                             # any rectangle, random fill density
        r0, r1 = sorted((ri(0, pn - 1), ri(0, pn - 1)))
        c0, c1 = sorted((ri(0, pn - 1), ri(0, pn - 1)))
        blk = _rand_c64(gen, r1 - r0 + 1, c1 - c0 + 1)
        if ri(0, 1):
            blk = blk * (torch.rand(blk.shape, generator=gen) < 0.3)
        p[r0:r1 + 1, c0:c1 + 1] = blk
        return p
    # disk family: the reference's r <= 1 support (|k| <= pn/4), optionally with samples around the natural box's edges
    yy, xx = torch.meshgrid(torch.arange(pn) - c, torch.arange(pn) - c, indexing="ij")
    inside = (yy * yy + xx * xx) <= h * h
    p = torch.where(inside, _rand_c64(gen, pn, pn), p)
    if kind == "disk_rim":                                                        # stays inside |k| <= pn/4
        n = ri(1, min(2 * h + 1, 140))
        a = ri(c - h, c + h - n + 1) if 2 * h + 1 >= n else c - h
        if ri(0, 1):
            p[a:a + n, c + (h if ri(0, 1) else -h)] = _rand_c64(gen, n)
        else:
            p[c + (h if ri(0, 1) else -h), a:a + n] = _rand_c64(gen, n)
        if ri(0, 3) == 0:
            p[c + (h if ri(0, 1) else -h), c + (h if ri(0, 1) else -h)] = 1j     # a corner
    if kind == "disk_junk":                                                       # a few samples within +-2 of the box edges
        for _ in range(ri(1, 4)):
            k = h + ri(-2, 2)
            t = ri(-h, h)
            pos = [(c + t, c + k), (c + t, c - k), (c + k, c + t), (c - k, c + t)][ri(0, 3)]
            if 0 <= pos[0] < pn and 0 <= pos[1] < pn:
                p[pos] = complex(_rand_c64(gen, 1)[0])
    return p


def _fuzz_case(seed):
    gen = torch.Generator().manual_seed(900000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))          # noqa: E731
    pick = lambda seq: seq[ri(0, len(seq) - 1)]                                      # noqa: E731
    pn = pick([64, 96, 128, 256, 256, 256, 512, 512])
    ps = pick([25, 25, 25, 10, 48])
    kind = pick(["disk", "disk", "disk", "disk_rim", "disk_rim", "disk_junk", "disk_junk", "box", "box", "single", "full", "empty"])
    planes = pick([1, 1, 1, 2, 3, 5])
    budget = {64: 40, 96: 30, 128: 24, 256: 12, 512: 5}[pn]                          # source points x planes the oracle gets
    S = max(1, min(ri(1, 14), budget // planes))
    c, h = pn // 2, pn // 4
    mode = pick(["narrow", "narrow", "narrow", "wide", "wide", "wrap"])
    lim = {"narrow": max(1, int(0.2 * pn)), "wide": c - h, "wrap": c}[mode]
    sh = torch.randint(-lim, lim + (0 if mode == "wrap" else 1), (S, 2), generator=gen, dtype=torch.int32)
    if S > 2 and ri(0, 1):
        sh[ri(0, S - 1)] = sh[0]                                                     # duplicates are legal source lists
    opts = {"poison": 1, "coarse": pick([0, 1, 2, 2])}
    for name, values in (("batch", [0, 0, 1, 2, 3, 5, 8]), ("groups", [0, 0, 1, 2, 3, 4]), ("xchunk", [0, 0, 1, 2, 3]),
                         ("tile", [0, 0, 4, 8]), ("plane_chunk", [0, 1, 2, 4]), ("gcombine", [1, 1, 0]),
                         ("rect", [1, 1, 0]), ("w64", [1, 1, 1, 0]), ("xrect", [1, 1, 0, 2]), ("force_generic", [0, 0, 0, 1]),
                         ("force_general", [0, 0, 0, 0, 1]), ("split", [1, 2, 2, 0]), ("embed", [1, 1, 0])):
        opts[name] = pick(values)
    pupils = torch.stack([_fuzz_pupil(gen, kind, pn) for _ in range(planes)])
    mft = torch.complex(torch.randn(pn, pn, generator=gen), torch.randn(pn, pn, generator=gen))
    prefill = torch.rand(planes, pn, pn, generator=gen) if ri(0, 1) else torch.zeros(planes, pn, pn)
    return dict(seed=seed, pn=pn, ps=ps, kind=kind, planes=planes, shifts=sh, opts=opts, pupils=pupils, mft=mft,
                prefill=prefill, mode=mode)


def _run_fuzz(L, dev, cases, what):
    """Every case through the engine with its random options under NaN-poisoned scratch, against the float64 closed form; returns
    the evaluation paths and kernel names reached."""
    o = O()
    failures, worst, paths, kernels = [], 0.0, {}, set()
    for cs in cases:
        pn, seed = cs["pn"], cs["seed"]
        _, N = L.Mask.calculateEpsilonN(None, 4 / pn, cs["ps"], WL)
        ref = torch.stack([o.abbe_raw_f64(cs["mft"], cs["pupils"][k], cs["shifts"], N) for k in range(cs["planes"])])
        scale = float(ref.max()) if float(ref.max()) > 0 else 1.0
        pre = (cs["prefill"].double() * 0.25 * scale).float()
        want = pre.double() + ref
        stacked = cs["planes"] > 1
        out = (pre if stacked else pre[0]).clone().to(dev)
        pup = (cs["pupils"] if stacked else cs["pupils"][0]).to(dev)
        tag = f"seed {seed}: pn {pn} N {N} {cs['kind']} planes {cs['planes']} S {cs['shifts'].shape[0]} {cs['mode']} {cs['opts']}"
        try:
            got = L.abbeIntensity(cs["mft"].to(dev), pup, cs["shifts"].to(dev), N, out=out, options=cs["opts"]).cpu().double()
        except Exception as exc:                                                     # noqa: BLE001  report every seed
            failures.append(f"{tag}: raised {type(exc).__name__}: {exc}")
            continue
        got = got if stacked else got[None]
        plan = nat().last_plan()
        key = ("general" if plan["general"] else "coarse" if plan["coarse_grid"] else "direct",
               "wave" if plan["wave_ypass"] else "r16", plan["variant"])
        if cs["kind"] != "empty":
            paths[key] = paths.get(key, 0) + 1
            kernels.update(nat().last_kernels())
        if not torch.isfinite(got).all():
            failures.append(f"{tag}: non-finite pixels (poisoned scratch read?) plan {plan}")
            continue
        e = float((got - want).abs().max() / scale)
        l2 = float(torch.linalg.norm(got - want) / max(float(torch.linalg.norm(want)), 1e-300))
        worst = max(worst, e)
        if e >= TOL_IMAGE_MAX or l2 >= TOL_IMAGE_L2:
            failures.append(f"{tag}: rel-to-max {e:.2e} rel-L2 {l2:.2e} plan {plan} kernels {nat().last_kernels()}")
    print(f"{what}: {len(cases)} cases, worst rel-to-max {worst:.2e}; evaluation paths hit: "
          + ", ".join(f"{k}: {v}" for k, v in sorted(paths.items(), key=str)) + f"; kernels: {sorted(kernels)}")
    assert not failures, "\n".join(failures[:40])
    return paths, kernels


def test_seeded_fuzz_against_the_float64_oracle(L, dev):
    paths, _ = _run_fuzz(L, dev, [_fuzz_case(seed) for seed in range(FUZZ_CASES)], "fuzz")
    # the fuzz is only worth its name if it reaches every family of the planner
    fams = {k[0] for k in paths}
    assert {"general", "coarse", "direct"} <= fams and any(k[1] == "wave" for k in paths) and any(k[1] == "r16" for k in paths), paths


# ------------------------------------------------------------------ the same at the sizes of the BASELINE GPU configurations
# (round-4 review, weak 1b: the small-size fuzz never reaches k_ypass_rect<10/11>, k_xpass_abbe<10..12>, k_ypass_coop*,
# k_xpass_split<13>, k_ypass_pair -- the kernels that carry configs 2-5.)  Same pupil families, random options incl. the
# 4096-only ones, pre-filled `out`, NaN-poisoned scratch, 1-2 source points x 1-3 planes, against the float64 closed form
# (O(pn^3) per point: the host's zgemm does a 2048^2 point in a fraction of a second).
def _fuzz_case_mid(seed, sizes, coarse=None, mode=None, kind=None, fixed=None):
    gen = torch.Generator().manual_seed(7700000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))          # noqa: E731
    pick = lambda seq: seq[ri(0, len(seq) - 1)]                                      # noqa: E731
    pn = pick(sizes)
    kind = pick(["disk", "disk", "disk", "disk", "disk_rim", "disk_rim", "disk_junk", "disk_junk", "box", "single"]) if kind is None else kind
    planes = pick([1, 1, 1, 2, 3]) if pn < 4096 else pick([1, 1, 2])
    S = pick([1, 2, 2]) if pn < 4096 else 1 if planes > 1 else pick([1, 2])
    c, h = pn // 2, pn // 4
    mode = mode or pick(["narrow", "narrow", "narrow", "wide", "wide", "wrap"])
    lim = {"narrow": max(1, int(0.2 * pn)), "wide": c - h, "wrap": c}[mode]
    sh = torch.randint(-lim, lim + (0 if mode == "wrap" else 1), (S, 2), generator=gen, dtype=torch.int32)
    opts = {"poison": 1, "coarse": pick([0, 2, 2, 2, 1]) if coarse is None else coarse}
    for name, values in (("batch", [0, 0, 1, 2, 3]), ("groups", [0, 0, 0, 1, 2, 4]), ("xchunk", [0, 0, 1, 2]),
                         ("tile", [0, 0, 0, 4, 8] + ([16] if pn == 4096 else [])), ("plane_chunk", [0, 1, 2, 4]), ("gcombine", [1, 1, 0]),
                         ("rect", [1, 1, 1, 0]), ("w64", [1, 1, 1, 1, 0]), ("xrect", [1, 1, 0, 2]), ("force_generic", [0, 0, 0, 0, 1]),
                         ("force_general", [0, 0, 0, 0, 0, 1]), ("split", [1, 2, 2, 0]), ("xsplit", [1, 1, 0]), ("w64_8192", [1, 1, 0]),
                         ("coopdma", [1, 1, 0]), ("rowpairs", [0, 0, 1])):
        opts[name] = pick(values)
    opts.update(fixed or {})
    pupils = torch.stack([_fuzz_pupil(gen, kind, pn) for _ in range(planes)])
    mft = torch.complex(torch.randn(pn, pn, generator=gen), torch.randn(pn, pn, generator=gen))
    prefill = torch.rand(planes, pn, pn, generator=gen) if ri(0, 1) else torch.zeros(planes, pn, pn)
    return dict(seed=seed, pn=pn, ps=25, kind=kind, planes=planes, shifts=sh, opts=opts, pupils=pupils, mft=mft,
                prefill=prefill, mode=mode)


def test_seeded_fuzz_mid_sizes_against_the_float64_oracle(L, dev):
    cases = [_fuzz_case_mid(seed, [1024, 1024, 2048]) for seed in range(48)]
    # the random options and pupil families leave the coarse grid most of the time (no wave kernels, generic kernels, long rims,
    # junk beyond the natural box): twelve more that stay on it -- disks, every launch-geometry option still random
    keep = {"w64": 1, "rect": 1, "force_generic": 0, "force_general": 0, "tile": 0}
    cases += [_fuzz_case_mid(500 + seed, [1024, 2048], coarse=2, mode=("narrow", "wide")[seed & 1], kind="disk", fixed=keep) for seed in range(12)]
    paths, kernels = _run_fuzz(L, dev, cases, "fuzz 1024/2048")
    fams = {k[0] for k in paths}
    assert {"coarse", "direct"} <= fams and paths.get(("coarse", "wave", 1), 0) >= 16, paths
    # the kernels of BASELINE configs 2, 3 and 5 (coarse grid) and their direct-path counterparts must have been reached
    for must in ("k_ypass_rect<10, 8, true", "k_ypass_rect<11, 8, true", "k_xpass_abbe<10, 0, true", "k_xpass_abbe<11, 0, true"):
        assert any(k.startswith(must) for k in kernels), (must, sorted(kernels))


def test_seeded_fuzz_4096_against_the_float64_oracle(L, dev):
    cases = [_fuzz_case_mid(1000 + seed, [4096]) for seed in range(10)]
    # ... and four on the DIRECT path with non-wrapping shifts (N = 8192: k_xpass_split + k_ypass_pair, or their radix-16
    # counterparts when the options say so), which the ten random draws above happen not to reach
    cases += [_fuzz_case_mid(1100 + seed, [4096], coarse=0, mode=("narrow", "wide")[seed & 1]) for seed in range(4)]
    # ... and six of which the disks STAY on the coarse grid (the random options and pupil families above leave it nine times out of ten: no
    # wave kernels, generic kernels, rims longer than 128 samples): disk / junk pupils, everything else random -- batch, groups,
    # chunk, planes in flight, the loader of the cooperative y-pass, tile 16 or automatic
    keep = {"w64": 1, "force_generic": 0, "force_general": 0, "rowpairs": 0}
    cases += [_fuzz_case_mid(1200 + seed, [4096], coarse=2, mode=("narrow", "wide")[seed & 1], kind=("disk", "disk", "disk_junk")[seed % 3],
                             fixed=dict(keep, tile=(0, 16)[seed % 2])) for seed in range(6)]
    paths, kernels = _run_fuzz(L, dev, cases, "fuzz 4096")
    assert paths.get(("coarse", "wave", 1), 0) >= 4 and any(k[0] == "direct" for k in paths), paths
    # config 4's kernels: the coarse grid's cooperative y-pass (either loader) and the 4096-point x-pass; the direct path's
    # 8192-point kernels
    assert any(k.startswith("k_ypass_coop") for k in kernels) and any(k.startswith("k_xpass_abbe<12, 0, true") for k in kernels), sorted(kernels)
    assert any(k.startswith(("k_xpass_split<13>", "k_ypass_pair<13", "k_ypass_acc<13", "k_xpass_abbe<13")) for k in kernels), sorted(kernels)


def test_plan_words_arrive_by_both_read_back_routes():
    """The planning read-back (round 6): the 14 plan words normally arrive in the calling thread's pinned, device-MAPPED buffer,
    published by k_plan_finish under a sequence flag the host polls; when the buffer cannot be mapped
    (LITHO_ABBE_NO_MAPPED_PLAN=1 emulates that) they come through the copy + stream wait.  Both routes in fresh processes:
    the same plan (box, batch, coarse grid, source count) and bit-identical images for a centred and a shifted (split) source,
    a stack, a single source point, and ten calls in a row (the sequence flag must never be taken for the previous call's)."""
    import json
    import os
    import subprocess
    import sys
    from helpers import ROOT
    code = r"""
import hashlib, json, sys, torch
sys.path.insert(0, %r)
import lithographysimulator_amd as L
from lithographysimulator_amd import _native as nat
from lithographysimulator_amd.synthetic import bernoulli_mask
dev = torch.device("cuda", 0)
out = []
def digest(t): return hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:16]
for pn, sx, planes in ((256, 0.0, 1), (512, 0.25, 1), (256, 0.0, 3)):
    mask = L.Mask(bernoulli_mask(pn), 25, dev)
    mft = mask.fraunhofer(193., True)
    bm = (L.LightSource(0.0, 0.5, pn, 0.7, device=dev) if sx == 0.0 else L.LightSource(0.4, 0.8, pn, 0.7, sx, -0.5, dev)).generateAnnular()
    ab = torch.tensor([0, 0, 0.01, 0, 60, 0.01], dtype=torch.float16)
    pf = (L.Pupil(pn, 193., 0.7, ab, dev).generatePupilFunction() if planes == 1 else
          L.throughFocusPupils(pn, 193., 0.7, ab, [-50.0, 0.0, 70.0], dev))
    for rep in range(10 if pn == 256 and planes == 1 else 1):
        img = L.abbeImage(mask, mft, pf, bm, 25, mask.deltaK, 193., True, dev)
        out.append([digest(img), {k: int(v) for k, v in nat.last_plan().items()}])
    sh = L.sourceShifts(bm, pn)
    one = L.abbeIntensity(mft, pf, sh[7:8], 2 * pn)
    out.append([digest(one), {k: int(v) for k, v in nat.last_plan().items()}])
print(json.dumps(out))
""" % ROOT
    res = {}
    for route in ("mapped", "copy"):
        env = dict(os.environ)
        env.pop("LITHO_ABBE_NO_MAPPED_PLAN", None)
        if route == "copy":
            env["LITHO_ABBE_NO_MAPPED_PLAN"] = "1"
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stderr[-1500:]
        res[route] = json.loads([l for l in r.stdout.splitlines() if l.startswith("[")][-1])
    assert res["mapped"] == res["copy"]
    plans = [p for _, p in res["mapped"]]
    assert plans[0]["coarse_grid"] == 1 and plans[0]["box_rows"] == 129 and len({d for d, _ in res["mapped"][:10]}) == 1
    assert any(p["planned_from_record"] == 2 for p in plans)               # the shifted source was split
