"""Host-side mirror of the reference object API: construction, fallbacks, and the loud
failure when no HIP device is present (there is no CPU fallback).  CPU only."""
import hashlib
import os
import re

import pytest
import torch

from helpers import ROOT, WL

import lithographysimulator_amd as L
from lithographysimulator_amd.synthetic import bernoulli_mask, lines_mask

CPU = torch.device("cpu")


def test_mask_demo_fallback_like_reference(capsys):
    m = L.Mask(device=CPU)                                   # mask.py:20-27: invalid -> 64x64 demo, never raises
    assert "Using demo instead" in capsys.readouterr().out
    assert m.pixelNumber == 64 and m.geometry.dtype == torch.int16 and m.deltaK == 4 / 64
    assert torch.equal(m.geometry, lines_mask(64))
    m2 = L.Mask(torch.ones(3, 5), 25, CPU)                   # non-square -> demo
    assert m2.pixelNumber == 64
    m3 = L.Mask(torch.ones(32, 32), 10, CPU)
    assert m3.pixelNumber == 32 and m3.pixelSize == 10 and m3.geometry.dtype == torch.int16


def test_calculate_epsilon_n_bound_and_unbound():
    m = L.Mask(bernoulli_mask(2048), 25, CPU)
    assert m.calculateEpsilonN(m.deltaK, 25, WL) == (1.0362694300518134, 4096)
    assert L.Mask.calculateEpsilonN(self=m, deltaK=m.deltaK, pixelSize=25, wavelength=WL)[1] == 4096   # imageformation.py:50
    assert m._nearest2SqInt(3000.0) == 2048 and m._nearest2SqInt(3.0) == 2                            # first minimum on ties


def test_no_cpu_fallback():
    m = L.Mask(bernoulli_mask(64), 25, CPU)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.fraunhofer(WL, True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.LightSource(0.4, 0.8, 64, device=CPU).generateAnnular()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.Pupil(64, WL, 0.7, None, CPU).generatePupilFunction()
    z = torch.zeros(64, 64, dtype=torch.complex64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.calculateFFTAerial(z, z, 64, 128)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.abbeImage(m, z, z, torch.zeros(64, 64, dtype=torch.int64), 25, m.deltaK, WL, True, CPU)


def test_mismatched_shapes_are_rejected_before_any_pointer_is_taken():
    """A default Pupil()/LightSource() is 64^2 (pupil.py:6, lightsource.py:5); with a 256^2 mask the reference dies
    with a broadcasting RuntimeError at imageformation.py:34.  Here the same call must raise (ValueError and
    RuntimeError both match) instead of handing a 64^2 buffer to kernels sized for 256^2."""
    from lithographysimulator_amd.imageformation import ShapeError
    m256 = torch.zeros(256, 256, dtype=torch.complex64)
    p64 = torch.zeros(64, 64, dtype=torch.complex64)
    sh = torch.zeros(3, 2, dtype=torch.int32)
    for exc in (ShapeError, ValueError, RuntimeError):
        with pytest.raises(exc, match="pupilF must be"):
            L.abbeIntensity(m256, p64, sh, 512)
    with pytest.raises(ShapeError, match="pupilF must be"):
        L.abbeIntensity(m256, torch.zeros(2, 64, 64, dtype=torch.complex64), sh, 512)
    with pytest.raises(ShapeError, match="square"):
        L.abbeIntensity(torch.zeros(256, 128, dtype=torch.complex64), p64, sh, 512)
    with pytest.raises(ShapeError, match="shifts must be"):
        L.abbeIntensity(m256, m256, torch.zeros(6, dtype=torch.int32), 512)
    with pytest.raises(ShapeError, match="calculateFFTAerial"):
        L.calculateFFTAerial(p64, m256, 256, 512)
    with pytest.raises(ShapeError, match="calculateFFTAerial"):
        L.calculateFFTAerial(m256, m256, 64, 128)
    mask = L.Mask(bernoulli_mask(256), 25, CPU)
    with pytest.raises(ValueError):            # source bitmap of the wrong size or wrong pupil
        L.abbeImage(mask, m256, p64, torch.zeros(256, 256, dtype=torch.int64), 25, mask.deltaK, WL, True, CPU)


def test_direct_solver_is_declared_out_of_scope():
    m = L.Mask(bernoulli_mask(64), 25, CPU)
    with pytest.raises(NotImplementedError):
        m.fraunhofer(WL, False)
    z = torch.zeros(64, 64, dtype=torch.complex64)
    with pytest.raises(NotImplementedError):
        L.abbeImage(m, z, z, z, 25, m.deltaK, WL, False, CPU)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from lithographysimulator_amd import _native as nat
    monkeypatch.setattr(nat, "_lib", None)
    monkeypatch.setattr(nat, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        nat.lib()


def test_osa_indices_and_pupil_defaults(capsys):
    assert [L.OSAindexToMN(j) for j in range(10)] == [(0, 0), (-1, 1), (1, 1), (-2, 2), (0, 2), (2, 2), (-3, 3),
                                                      (-1, 3), (1, 3), (3, 3)]
    p = L.Pupil(device=CPU)
    assert "Assuming perfect system" in capsys.readouterr().out
    assert p.aberrations.dtype == torch.float16 and p.aberrations.tolist() == [0.0]
    ls = L.LightSource(device=CPU)
    assert (ls.sigmaInner, ls.sigmaOuter, ls.pixelNumber, ls.NA) == (0, 0.6, 64, 0.7)      # lightsource.py:5


def test_embedded_size_rule():
    """Which grid a pn x pn problem runs at (litho_abbe_embedded_size, host-only): N and N / 2 as they are, everything else even
    in the next of the two -- N / 2 from 256 up (the coarse grid applies), N from 1024 up."""
    table = {(1024, 2048): 1024, (2048, 2048): 2048, (1000, 2048): 1024, (1500, 2048): 2048, (2000, 4096): 2048, (3000, 4096): 4096,
             (768, 1024): 1024, (200, 512): 256, (300, 512): 300, (300, 1024): 512, (96, 128): 96, (100, 256): 100,
             (1024, 4096): 2048, (256, 1024): 512, (64, 256): 64, (2000, 2048): 2048, (8192, 16384): 8192}
    for (pn, N), want in table.items():
        assert L.embeddedSize(pn, N) == want, (pn, N)
        pe = L.embeddedSize(pn, N)
        assert pe == pn or ((pe - pn) % 2 == 0 and pe in (N, N // 2) and pe > pn)
    with pytest.raises(ValueError):
        L.embeddedSize(1001, 2048)                     # odd sizes are not served at all


def test_synthetic_masks_are_reproducible():
    h = hashlib.sha256(bernoulli_mask(256).numpy().tobytes()).hexdigest()
    assert h == hashlib.sha256(bernoulli_mask(256).numpy().tobytes()).hexdigest()
    b = bernoulli_mask(1024)
    assert 0.49 < float(b.float().mean()) < 0.51 and b.dtype == torch.int16
    assert int(lines_mask(128).sum()) == 4 * (46 * 2) * (4 * 2)
    with pytest.raises(ValueError):
        lines_mask(100)


def test_product_never_touches_the_oracle():
    """The oracle is the checker: nothing under lithographysimulator_amd/ may import, call, link or
    execute anything from oracle/ (nor read /root/reference)."""
    pkg = os.path.join(ROOT, "lithographysimulator_amd")
    bad = []
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                continue
            text = open(os.path.join(dirpath, f)).read()
            if re.search(r"\boracle\b|/root/reference|abbe_ref", text):
                bad.append(os.path.join(dirpath, f))
    assert not bad, bad
    mk = open(os.path.join(ROOT, "Makefile")).read()
    assert "libabbe_ref" not in mk.split("$(OUT):")[1].split("\n\n")[0]      # not linked into the product
