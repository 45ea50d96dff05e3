"""Layout import on the GPU (`-m gpu`): the device rasteriser against its CPU restatement bit for bit, and
GDSII -> Mask -> abbeImage end to end.  (No reference counterpart: README.md:20-22 lists GDSII import as a goal.)"""
import numpy as np
import pytest
import torch

from helpers import NA, PS, WL, f16

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the -m gpu tests need an MI355X"
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def L():
    import lithographysimulator_amd as L
    return L


def random_polygons(rng, n, span, kinds=("rect", "tri", "blob")):
    polys = []
    for _ in range(n):
        cx, cy = rng.uniform(-0.1 * span, 1.1 * span, 2)
        kind = kinds[rng.integers(len(kinds))]
        if kind == "rect":
            w, h = rng.uniform(0.01 * span, 0.3 * span, 2)
            q = np.array([[cx, cy], [cx + w, cy], [cx + w, cy + h], [cx, cy + h]])
        elif kind == "tri":
            q = np.array([cx, cy]) + rng.uniform(-0.2 * span, 0.2 * span, (3, 2))
        else:                                       # a star-shaped (possibly concave) polygon
            k = int(rng.integers(5, 12))
            ang = np.sort(rng.uniform(0, 2 * np.pi, k))
            rad = rng.uniform(0.02 * span, 0.2 * span, k)
            q = np.array([cx, cy]) + np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=1)
        if rng.integers(2):
            q = q[::-1]                             # either orientation
        polys.append(q)
    return polys


@pytest.mark.parametrize("pn,n,ps,seed", [(64, 12, 1.0, 0), (256, 60, 2.5, 1), (1024, 300, 25.0, 2), (2048, 40, 10.0, 3), (100, 30, 7.0, 4)])
def test_rasteriser_matches_cpu_restatement(L, dev, pn, n, ps, seed):
    from lithographysimulator_amd import layout as LY
    from oracle import layout_oracle as LO
    rng = np.random.default_rng(seed)
    polys = random_polygons(rng, n, pn * ps)
    # vertices ON pixel centres and on pixel boundaries, horizontal and vertical edges through centres
    polys.append(np.array([[2.5, 3.5], [10.5, 3.5], [10.5, 9.5], [2.5, 9.5]]) * ps)
    polys.append(np.array([[20.0, 20.0], [30.0, 20.0], [25.0, 28.0]]) * ps)
    x0, y0 = float(rng.uniform(-3, 3)) * ps, float(rng.uniform(-3, 3)) * ps
    got = L.rasterizeLayout(polys, pn, ps, origin=(x0, y0), device=dev)
    assert got.dtype == torch.int16 and tuple(got.shape) == (pn, pn)
    want = LO.rasterize_edges(LY.polygonEdges(polys), pn, x0, y0, ps)
    diff = int((got.cpu().numpy() != want).sum())
    print(f"raster {pn}^2, {len(polys)} polygons: {int(want.sum())} pixels set, {diff} differ")
    assert diff == 0
    assert 0 < want.sum() < pn * pn


def test_rasteriser_edge_cases(L, dev):
    from lithographysimulator_amd import _native as nat
    empty = L.rasterizeLayout([], 32, 5.0, origin=(0.0, 0.0), device=dev)
    assert int(empty.sum()) == 0
    full = L.rasterizeLayout([np.array([[-1e6, -1e6], [1e6, -1e6], [1e6, 1e6], [-1e6, 1e6]])], 33, 5.0, origin=(0.0, 0.0), device=dev)
    assert int(full.sum()) == 33 * 33                                  # pn need not be even or a power of two here
    # overlapping squares: the union, not the exclusive-or; a degenerate (zero-area) polygon adds nothing
    sq = lambda a, b: np.array([[a, a], [b, a], [b, b], [a, b]], dtype=float)
    u = L.rasterizeLayout([sq(0, 10), sq(5, 15)[::-1], np.array([[3.0, 3.0], [8.0, 8.0], [3.0, 3.0]])], 16, 1.0, origin=(0.0, 0.0), device=dev).cpu().numpy()
    want = np.zeros((16, 16), dtype=np.int16)
    want[0:10, 0:10] = 1
    want[5:15, 5:15] = 1
    assert np.array_equal(u, want)
    # default origin: the window is centred on the polygons' bounding box
    c = L.rasterizeLayout([sq(100, 140)], 8, 10.0, device=dev).cpu().numpy()
    want = np.zeros((8, 8), dtype=np.int16)
    want[2:6, 2:6] = 1
    assert np.array_equal(c, want)
    # non-finite coordinates (NaN, +-inf) are skipped edge by edge, exactly as the CPU restatement does
    from oracle import layout_oracle as LO
    from lithographysimulator_amd import layout as LY
    good = LY.polygonEdges([sq(2, 7)])
    bad = np.array([[np.inf, 0.0, 3.0, 5.0], [1.0, -np.inf, 1.0, 4.0], [np.nan, 0.0, 2.0, 9.0], [0.0, 0.0, -np.inf, 8.0], [4.0, 1.0, np.inf, 6.0]])
    ed = torch.from_numpy(np.concatenate([good, bad])).to(dev)
    geo = torch.zeros((16, 16), dtype=torch.int16, device=dev)
    work = torch.zeros(nat.rasterize_work_bytes(16), dtype=torch.uint8, device=dev)
    assert nat.lib().litho_rasterize_edges(nat.ptr(ed), ed.shape[0], 16, 0.0, 0.0, 1.0, nat.ptr(work), work.numel(), nat.ptr(geo), nat.stream_ptr(dev)) == 0
    assert np.array_equal(geo.cpu().numpy(), LO.rasterize_edges(np.concatenate([good, bad]), 16, 0.0, 0.0, 1.0))
    assert np.array_equal(geo.cpu().numpy(), LO.rasterize_edges(good, 16, 0.0, 0.0, 1.0))
    # argument checks of the C entry
    geo = torch.zeros((8, 8), dtype=torch.int16, device=dev)
    work = torch.zeros(nat.rasterize_work_bytes(8), dtype=torch.uint8, device=dev)
    e = torch.zeros(4, dtype=torch.float64, device=dev)
    call = lambda *a: nat.lib().litho_rasterize_edges(*a)
    assert call(nat.ptr(e), 1, 8, 0.0, 0.0, 0.0, nat.ptr(work), work.numel(), nat.ptr(geo), nat.stream_ptr(dev)) == -1       # pixel 0
    assert call(nat.ptr(e), 1, 8, 0.0, 0.0, 1.0, nat.ptr(work), work.numel() - 1, nat.ptr(geo), nat.stream_ptr(dev)) == -3   # workspace
    assert call(None, 1, 8, 0.0, 0.0, 1.0, nat.ptr(work), work.numel(), nat.ptr(geo), nat.stream_ptr(dev)) == -1
    assert call(None, 0, 8, 0.0, 0.0, 1.0, nat.ptr(work), work.numel(), nat.ptr(geo), nat.stream_ptr(dev)) == 0
    assert nat.rasterize_work_bytes(8) == 8 * 9 * 4


def test_gdsii_to_image_end_to_end(L, dev, tmp_path):
    """A GDSII file with a cell of two bars placed by an AREF (a line/space grating) and a mirrored SREF -> Mask ->
    aerial image.  The mask equals the raster built by hand, and so does the image."""
    from lithographysimulator_amd import layout as LY
    pn = 256
    lib = LY.GdsLibrary("GRATING", 1e-3, 1e-9)
    bar = LY.GdsStructure("BAR")
    bar.elements.append(LY.GdsElement("boundary", layer=7, datatype=0, xy=np.array([[0, 0], [100, 0], [100, 3200], [0, 3200], [0, 0]])))
    bar.elements.append(LY.GdsElement("boundary", layer=9, datatype=0, xy=np.array([[0, 0], [5000, 0], [5000, 5000], [0, 5000], [0, 0]])))   # another layer: ignored
    top = LY.GdsStructure("TOP")
    top.elements.append(LY.GdsElement("aref", sname="BAR", xy=np.array([[1000, 1600], [1000 + 12 * 350, 1600], [1000, 1600 + 3200]]), cols=12, rows=1))
    top.elements.append(LY.GdsElement("path", layer=7, datatype=0, width=150, pathtype=0, xy=np.array([[1000, 1000], [5200, 1000]])))
    lib.structures["BAR"], lib.structures["TOP"] = bar, top
    path = tmp_path / "grating.gds"
    LY.writeGDSII(lib, str(path))
    mask = L.maskFromGDSII(str(path), pn, PS, top="TOP", layers=[(7, 0)], origin=(0.0, 0.0), device=dev)
    assert isinstance(mask, L.Mask) and mask.pixelNumber == pn and mask.pixelSize == PS
    hand = torch.zeros((pn, pn), dtype=torch.int16)
    for i in range(12):
        hand[64:192, (1000 + 350 * i) // 25:(1000 + 350 * i + 100) // 25] = 1          # 100 nm bars on a 350 nm pitch: 4 px of 14
    hand[37:43, 40:208] = 1                                                            # the 150 nm wire: rows 925 .. 1075 nm
    assert torch.equal(mask.geometry.cpu(), hand)
    src = L.LightSource(0.0, 0.6, pn, NA, device=dev).generateAnnular()
    pup = L.Pupil(pn, WL, NA, f16([0, 0, 0, 0, 40]), dev).generatePupilFunction()
    img = L.abbeImage(mask, mask.fraunhofer(WL, True), pup, src, PS, mask.deltaK, WL, True, dev).cpu()
    hm = L.Mask(hand, PS, dev)
    ref = L.abbeImage(hm, hm.fraunhofer(WL, True), pup, src, PS, hm.deltaK, WL, True, dev).cpu()
    assert torch.equal(img, ref) and img.shape[0] >= pn - 2 and float(img.max()) > 0
    # and from the library object, default origin (centred on the layout), every layer
    m2 = L.maskFromGDSII(lib, pn, PS, device=dev)
    assert int(m2.geometry.sum()) > int(mask.geometry.sum())
