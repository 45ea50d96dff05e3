"""CPU oracle for the Abbe aerial-image hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  ``lithographysimulator_amd`` never does (tests/test_host_logic_cpu.py,
``test_product_never_touches_the_oracle``, enforces that).

It is a restatement, in our own words, of what quarterwave0/LithographySimulator
computes on its PyTorch-CPU path.  The arithmetic library the reference uses is
``torch`` (not vendored, no version pinned by the reference; this image has
torch 2.10.0+rocm7.0, and the fp16 scalar-promotion behaviour below is that
version's).  Every function cites the reference file:line it follows.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real
reference in the build container and stores its outputs under
``tests/golden/*.npz``; ``tests/test_oracle_vs_golden.py`` checks every function
here against them (bit-exact for source bitmaps, fp16 wavefronts and pupils;
fp32 tolerance for fields and images).

fp16 convention used throughout: ``h(x)`` rounds an fp32 tensor to fp16
(round-to-nearest-even) and widens back.  The reference keeps its sigma grids,
radii, angles and Zernike sums in fp16 tensors; torch-CPU evaluates each fp16
op in fp32 and rounds the result once, which is what the explicit ``h`` calls
reproduce.  Transcendentals are torch fp32 CPU ops (the same SLEEF kernels the
fp16 path widens into).
"""
from __future__ import annotations

import math
from typing import Iterable, Sequence, Tuple

import numpy as np
import torch

F32 = torch.float32
TWO_POWERS = (2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384)


def h(x: torch.Tensor) -> torch.Tensor:
    """fp32 -> fp16 (RNE) -> fp32."""
    return x.to(torch.float16).to(F32)


def _hs(v: float) -> torch.Tensor:
    """Python scalar rounded to fp16, as torch does for add/remainder/compare
    between an fp16 tensor and a Python number."""
    return h(torch.tensor(float(v), dtype=F32))


def _fs(v: float) -> torch.Tensor:
    """Python scalar as fp32 (what torch keeps for fp16-tensor * scalar)."""
    return torch.tensor(float(v), dtype=F32)


# --------------------------------------------------------------------------
# a6  FFT sizing  (mask.py:63-72)
# --------------------------------------------------------------------------
def calculate_epsilon_n(deltaK: float, pixelSize: float, wavelength: float) -> Tuple[float, int]:
    """beta = 1/((deltaK*pixelSize)/wavelength); N = power of two nearest beta
    (first minimum wins; the distance is evaluated in fp32 because the
    reference subtracts a Python float from an int16 tensor, mask.py:64-65);
    epsilon = N/beta (mask.py:68-70)."""
    beta = ((deltaK * pixelSize) / wavelength) ** -1
    sq = np.asarray(TWO_POWERS, dtype=np.float32)
    N = int(TWO_POWERS[int(np.argmin(np.abs(sq - np.float32(beta))))])
    return N / beta, N


# --------------------------------------------------------------------------
# a5  source sampling  (lightsource.py:34-73)
# --------------------------------------------------------------------------
def _sigma_axis(pn: int, shift: float) -> torch.Tensor:
    """torch.arange(-2-shift, 2-shift, 4/pn, dtype=fp16) (lightsource.py:39-40, pupil.py:53).
    torch-CPU fills fp16 aranges 16 lanes at a time: the chunk base start+step*i0 is
    evaluated in fp32 and ROUNDED TO fp16, then lane l is h(base + l*step); a tail shorter
    than 16 is h(start + step*i).  Identical to h(start+i*step) whenever the shift is
    fp16-exact (every BASELINE config), differs for shifts such as 0.2.  The element count
    is ceil((end-start)/step) in double."""
    start, end, step = -2.0 - shift, 2.0 - shift, 4.0 / pn
    n = int(math.ceil((end - start) / step))
    i = torch.arange(n)
    fs, fst = _fs(start), _fs(step)
    i0 = (i // 16) * 16
    vec = h(h(fs + fst * i0.to(F32)) + (i - i0).to(F32) * fst)
    tail = h(fs + fst * i.to(F32))
    return torch.where(i < (n // 16) * 16, vec, tail)


def _source_radius(pn: int, shiftX: float, shiftY: float):
    sx = _sigma_axis(pn, shiftX)
    sy = _sigma_axis(pn, shiftY)
    X = sx[None, :].expand(sy.numel(), sx.numel())   # indexing='xy': X varies along columns
    Y = sy[:, None].expand(sy.numel(), sx.numel())
    O = h(torch.sqrt(h(h(X * X) + h(Y * Y))))          # lightsource.py:47 / :61
    return X, Y, O


def source_annular(sigmaIn: float, sigmaOut: float, pn: int,
                   shiftX: float = 0.0, shiftY: float = 0.0) -> torch.Tensor:
    """lightsource.py:34-50.  int64 0/1 bitmap; thresholds compared in fp16."""
    _, _, O = _source_radius(pn, shiftX, shiftY)
    lit = (O >= _hs(sigmaIn)) & (O <= _hs(sigmaOut))
    return lit.to(torch.int64)


def source_quasar(sigmaIn: float, sigmaOut: float, pn: int, count: int, rotation: float,
                  shiftX: float = 0.0, shiftY: float = 0.0) -> torch.Tensor:
    """lightsource.py:52-73: annulus with ``count`` open angular wedges removed."""
    X, Y, O = _source_radius(pn, shiftX, shiftY)
    theta = h(h(torch.atan2(Y, X)) + _hs(rotation))                 # :62
    theta = h(torch.remainder(theta, _hs(2 * math.pi)))             # :63
    lit = (O >= _hs(sigmaIn)) & (O <= _hs(sigmaOut))                # :65
    spacing = math.pi / count                                       # :67
    for gap in range(count):                                        # :70-71
        lo, hi = _hs((gap + gap) * spacing), _hs((gap + gap + 1) * spacing)
        lit = lit & ~((lo < theta) & (theta < hi))
    return lit.to(torch.int64)


def source_shifts(bitmap: torch.Tensor, pn: int) -> torch.Tensor:
    """imageformation.py:59: (argwhere(ls) - pn//2) as int32 [S,2], row-major."""
    return (torch.argwhere(bitmap) - (pn // 2)).to(torch.int32)


# --------------------------------------------------------------------------
# a4  pupil  (pupil.py:46-111)
# --------------------------------------------------------------------------
def osa_index_to_mn(j: int) -> Tuple[int, int]:
    """pupil.py:82-86."""
    n = math.ceil(0.5 * (-3 + math.sqrt(9 + 8 * j)))
    m = 2 * j - n * (n + 2)
    return m, n


def _pupil_grid(pn: int):
    x = _sigma_axis(pn, 0.0)                       # pupil.py:53
    X = x[None, :].expand(pn, pn)
    Y = x[:, None].expand(pn, pn)
    r = h(torch.sqrt(h(h(X * X) + h(Y * Y))))      # pupil.py:56
    theta = h(torch.atan2(Y, X))                   # pupil.py:57
    return r, theta


def zernike_term(m: int, n: int, pn: int, coeff16: torch.Tensor, grid=None) -> torch.Tensor:
    """pupil.py:46-77 (generateZ).  ``coeff16`` is an fp32 scalar tensor holding an
    fp16-representable value.  Returns fp32 holding fp16 values."""
    r, theta = grid if grid is not None else _pupil_grid(pn)
    lLim = int((n - abs(m)) / 2)
    ilLim = int((n + abs(m)) / 2)
    acc = torch.zeros_like(r)
    for k in range(lLim + 1):                      # :62-65
        static = ((-1) ** k * math.factorial(n - k)) / (
            math.factorial(k) * math.factorial(ilLim - k) * math.factorial(lLim - k))
        acc = acc + h(_fs(static) * h(torch.pow(r, float(n - 2 * k))))
    R = h(acc)                                      # torch.sum(dim=0): fp32 accumulate, one rounding (:67)
    Nmn = math.sqrt((2 * n + 1) / (1 + (1 if m == 0 else 0)))   # :68
    if m >= 0:                                      # :70-73
        cN = h(coeff16 * _fs(Nmn))
        trig = h(torch.cos(h(_fs(m) * theta)))
    else:
        cN = h(coeff16 * _fs(-Nmn))
        trig = h(torch.sin(h(_fs(m) * theta)))
    Z = h(h(cN * R) * trig)
    return torch.where(r <= 1, Z, torch.zeros_like(Z))          # :77


def scaled_aberrations(aberrations16: torch.Tensor, NA: float, wavelength: float) -> torch.Tensor:
    """pupil.py:91-92: coefficient 4 becomes c4*NA^2/(4*lambda), rounded to fp16 after the
    multiply and again after the divide.  len==4 raises IndexError like the reference (Q3)."""
    ab = h(aberrations16.to(F32)).clone()
    if len(ab) >= 4:
        if len(ab) == 4:
            raise IndexError("index 4 is out of bounds for dimension 0 with size 4")
        ab[4] = h(h(ab[4] * _fs(NA ** 2)) / _fs(4 * wavelength))
    return ab


def wavefront_error(aberrations16: torch.Tensor, pn: int, NA: float, wavelength: float) -> torch.Tensor:
    """pupil.py:88-100 on a FRESH coefficient vector (one rescale).  Returns fp16 tensor W."""
    ab = scaled_aberrations(aberrations16, NA, wavelength)
    grid = _pupil_grid(pn)
    W = torch.zeros((pn, pn), dtype=F32)
    for j in range(len(ab)):
        m, n = osa_index_to_mn(j)
        W = h(W + zernike_term(m, n, pn, ab[j], grid))
    return W.to(torch.float16)


def pupil_from_wavefront(W16: torch.Tensor, pn: int) -> torch.Tensor:
    """pupil.py:102-111: phi = exp(1j*2*pi*W) as complex64, zero where r>1."""
    WE = W16.to(torch.complex64)
    phi = torch.exp(1j * 2 * torch.pi * WE)
    r, _ = _pupil_grid(pn)
    return torch.where(r <= 1, phi, torch.zeros_like(phi))


def pupil_function(aberrations16, pn: int, NA: float, wavelength: float) -> torch.Tensor:
    """Pupil.generatePupilFunction (pupil.py:32-35); ``None`` = perfect lens (pupil.py:21-23)."""
    if aberrations16 is None:
        aberrations16 = torch.tensor([0], dtype=torch.float16)
    return pupil_from_wavefront(wavefront_error(aberrations16, pn, NA, wavelength), pn)


# --------------------------------------------------------------------------
# mask spectrum pre-step (mask.py:74-90)
# --------------------------------------------------------------------------
def bilinear_resize(img: torch.Tensor, scale: float) -> torch.Tensor:
    """F.interpolate(mode='bilinear', scale_factor=scale, align_corners=False) on a 2-D
    fp32 image: out size floor(in*scale); src = (dst+0.5)/scale - 0.5 clamped at 0; the
    given scale (not the size ratio) maps coordinates (mask.py:77, imageformation.py:71)."""
    n_in = img.shape[0]
    n_out = int(math.floor(n_in * scale))
    if n_out == n_in:        # torch: equal sizes are a plain copy whatever the scale factor says
        return img.clone()
    rs = torch.tensor(1.0 / scale, dtype=F32)       # torch keeps the reciprocal scale in fp32
    dst = torch.arange(n_out, dtype=torch.float64)
    # torch-CPU evaluates rs*(dst+0.5)-0.5 as ONE fused multiply-add (single rounding):
    # the fp32 x fp32 product is exact in double, so round-once == double then fp32.
    src = torch.clamp((rs.double() * (dst + 0.5) - 0.5).to(F32), min=0.0)
    i0 = src.floor().to(torch.int64)
    i1 = torch.clamp(i0 + 1, max=n_in - 1)
    l1 = src - i0.to(F32)
    l0 = 1.0 - l1
    # torch interpolates x inside y: out = l0y*(l0x*a + l1x*b) + l1y*(l0x*c + l1x*d)
    a = img[i0][:, i0]; b = img[i0][:, i1]; c = img[i1][:, i0]; d = img[i1][:, i1]
    out = l0[:, None] * (l0[None, :] * a + l1[None, :] * b) + l1[:, None] * (l0[None, :] * c + l1[None, :] * d)
    return out


def mask_spectrum(geometry: torch.Tensor, pixelSize: float, wavelength: float) -> torch.Tensor:
    """Mask.fraunhofer(wavelength, fft=True): mask.py:37-40 and 74-90."""
    pn = geometry.shape[0]
    eps, N = calculate_epsilon_n(4 / pn, pixelSize, wavelength)
    scaled = bilinear_resize(geometry.to(F32), eps)                 # :76-77
    pW = ((N - pn) - (scaled.shape[0] - pn)) // 2                   # :79
    corr = scaled.shape[0] % 2                                      # :80
    padded = torch.nn.functional.pad(scaled, (pW, pW + corr, pW, pW + corr))
    spec = torch.fft.ifftshift(torch.fft.fft2(torch.fft.fftshift(padded), norm="backward"))  # :83-85
    trim = (N - pn) // 2                                            # :87-88
    return torch.nn.functional.pad(spec, (-trim, -trim, -trim, -trim))


# --------------------------------------------------------------------------
# a2  per-source-point field  (imageformation.py:32-45)
# --------------------------------------------------------------------------
def field_opchain(pf: torch.Tensor, maskFT: torch.Tensor, pn: int, N: int) -> torch.Tensor:
    """The reference's op chain, op for op: multiply, zero-pad to N, fftshift,
    unnormalised inverse FFT, ifftshift, crop the centre pn x pn."""
    prod = pf * maskFT                                              # :34
    pW = (N - pn) // 2                                              # :36
    padded = torch.nn.functional.pad(prod, (pW, pW, pW, pW))        # :37
    out = torch.fft.ifftshift(torch.fft.ifft2(torch.fft.fftshift(padded), norm="forward"))  # :39-41
    return out[pW:pW + pn, pW:pW + pn]                              # :43


def centred_dft_matrix(pn: int, N: int, dtype=torch.complex128) -> torch.Tensor:
    """F[q,i] = exp(+2*pi*1j*(i-c)*(q-c)/N), c = pn//2: the closed form of the chain above
    for even N-pn."""
    c = pn // 2
    k = torch.arange(pn, dtype=torch.float64) - c
    ang = 2 * math.pi * torch.outer(k, k) / N
    return torch.complex(torch.cos(ang), torch.sin(ang)).to(dtype)


def field_closed_form(pupil: torch.Tensor, maskFT: torch.Tensor, dy: int, dx: int, N: int) -> torch.Tensor:
    """complex128 E = F A F^T with A[i,j] = P[(i-dy) mod pn,(j-dx) mod pn] * M[i,j]
    (imageformation.py:63 roll + :32-45)."""
    pn = maskFT.shape[0]
    A = torch.roll(pupil.to(torch.complex128), shifts=(int(dy), int(dx)), dims=(0, 1)) * maskFT.to(torch.complex128)
    F = centred_dft_matrix(pn, N)
    return F @ A @ F.T


# --------------------------------------------------------------------------
# a1  Abbe accumulation  (imageformation.py:54-67)  and  a3 post-process (:69-77)
# --------------------------------------------------------------------------
def abbe_raw(maskFT: torch.Tensor, pupil: torch.Tensor, shifts: torch.Tensor, N: int) -> torch.Tensor:
    """fp32 sum over source points of |E_s|^2, sequential in list order, exactly the
    reference's loop body (imageformation.py:62-67; the complex64 accumulator with a real
    addend is an fp32 accumulation, Q8).  This is also the CPU baseline bench.py times."""
    pn = maskFT.shape[0]
    if N < pn:
        raise RuntimeError(f"FFT size N={N} is smaller than the mask ({pn}); the reference fails here too (Q6)")
    image = torch.zeros((pn, pn), dtype=F32)
    sh = shifts.tolist()
    for dy, dx in sh:
        rolled = torch.roll(pupil, shifts=(dy, dx), dims=(0, 1))
        image += torch.abs(field_opchain(rolled, maskFT, pn, N)) ** 2
    return image


def abbe_raw_f64(maskFT: torch.Tensor, pupil: torch.Tensor, shifts: torch.Tensor, N: int) -> torch.Tensor:
    """Same sum in complex128 / float64 through the closed form (truth for tolerance
    budgeting; O(pn^3) per source point)."""
    pn = maskFT.shape[0]
    F = centred_dft_matrix(pn, N)
    M = maskFT.to(torch.complex128)
    P = pupil.to(torch.complex128)
    image = torch.zeros((pn, pn), dtype=torch.float64)
    for dy, dx in shifts.tolist():
        E = F @ (torch.roll(P, shifts=(dy, dx), dims=(0, 1)) * M) @ F.T
        image += E.real ** 2 + E.imag ** 2
    return image


def post_process(raw: torch.Tensor, epsilon: float) -> torch.Tensor:
    """imageformation.py:69-77: |.|, bilinear resample by 1/epsilon, zero-pad with
    pW=(pn-round(pn/eps))//2 before and pW+size%2 after (Q5: 4096 -> 4094)."""
    pn = raw.shape[0]
    img = bilinear_resize(torch.abs(raw).to(F32), 1.0 / epsilon)
    pW = (pn - round(pn / epsilon)) // 2
    corr = img.shape[0] % 2
    return torch.nn.functional.pad(img, (pW, pW + corr, pW, pW + corr))


def abbe_image(maskFT, pupil, bitmap, pixelSize: float, deltaK: float, wavelength: float) -> torch.Tensor:
    """abbeImage(..., fft=True) end to end (imageformation.py:47-77)."""
    eps, N = calculate_epsilon_n(deltaK, pixelSize, wavelength)
    pn = maskFT.shape[0]
    return post_process(abbe_raw(maskFT, pupil, source_shifts(bitmap, pn), N), eps)
