"""Pupil: same object API as the reference's pupil.py; the fp16 Zernike sum and the phase
exp(i 2 pi W) are evaluated by a HIP kernel."""
from math import ceil, sqrt

import numpy as np
import torch

from . import _native as nat


def diracd(v):
    return 1 if v == 0 else 0


def OSA(m, n):
    """pupil.py:79-80."""
    return (n * (n + 2) + m) / 2


def OSAindexToMN(ji):
    """pupil.py:82-86."""
    n = ceil(1 / 2 * (-3 + sqrt(9 + 8 * ji)))
    m = (2 * ji) - (n * (n + 2))
    return m, n


def _run_pupil(aberrations: torch.Tensor, pixelNumber, NA, wavelength, device, want_w, want_phi, rescale=True):
    dev = nat.require_gpu(device)
    pn = int(pixelNumber)
    J = len(aberrations)
    if J < 1:
        raise ValueError("aberrations must hold at least one coefficient")
    coeffs = aberrations.detach().to(torch.float16).cpu().contiguous().view(torch.int16).numpy().astype(np.uint16)
    coeffs = np.ascontiguousarray(coeffs)
    W = torch.empty((pn, pn), dtype=torch.float16, device=dev) if want_w else None
    phi = torch.empty((pn, pn), dtype=torch.complex64, device=dev) if want_phi else None
    with torch.cuda.device(dev):
        nat.check(nat.lib().litho_pupil(coeffs.ctypes.data, J, pn, float(NA), float(wavelength), 0 if rescale else 1,
                                        nat.ptr(W) if want_w else None, nat.ptr(phi) if want_phi else None,
                                        nat.stream_ptr(dev)), "litho_pupil")
    if rescale and J >= 5:
        # pupil.py:91-92 rescales the CALLER's tensor in place; keep that visible
        aberrations[4] = torch.from_numpy(coeffs[4:5].view(np.float16).copy())[0].to(aberrations.dtype)
    return W, phi


def generateWavefrontError(aberrations, pixelNumber, NA, wavelength, device):
    """pupil.py:88-100: complex64 [pn,pn] whose real part is the fp16 wavefront error W."""
    W, _ = _run_pupil(aberrations, pixelNumber, NA, wavelength, device, True, False)
    return W.type(torch.complex64)


def generateZ(m, n, pixelNumber, coeff, device):
    """pupil.py:46-77: one Zernike term (fp16 [pn,pn])."""
    j = int(OSA(m, n))
    ab = torch.zeros(j + 1, dtype=torch.float16)
    ab[j] = float(coeff)
    if j + 1 == 4:                       # a length-4 vector is rejected (Q3); pad, term 4 is zero
        ab = torch.cat([ab, torch.zeros(1, dtype=torch.float16)])
    W, _ = _run_pupil(ab, pixelNumber, 1.0, 1.0, device, True, False, rescale=False)
    return W


def generatePhi(WE, pixelNumber, device):
    """pupil.py:102-111: phi = exp(1j*2*pi*WE), zero where r > 1."""
    dev = nat.require_gpu(device)
    pn = int(pixelNumber)
    we = WE.to(device=dev, dtype=torch.complex64).contiguous()
    phi = torch.empty((pn, pn), dtype=torch.complex64, device=dev)
    with torch.cuda.device(dev):
        nat.check(nat.lib().litho_pupil_phase(nat.ptr(we), pn, nat.ptr(phi), nat.stream_ptr(dev)), "litho_pupil_phase")
    return phi


class Pupil:
    """Mirror of reference pupil.py:4-38."""

    def __init__(self, pixelNumber: int = 64, wavelength=193., NA=0.7, aberrations: torch.Tensor = None,
                 device: torch.device = None):
        self.device = nat.pick_device(device, "pupil function")            # pupil.py:8-19
        if aberrations is None:
            print("No aberrations defined for pupil function! Assuming perfect system.")
            self.aberrations = torch.tensor([0], dtype=torch.float16)      # pupil.py:21-23
        else:
            self.aberrations = aberrations                                  # kept by reference (Q2)
        self.pixelNumber = pixelNumber
        self.wavelength = wavelength
        self.NA = NA

    def generatePupilFunction(self) -> torch.Tensor:
        """pupil.py:32-35."""
        _, phi = _run_pupil(self.aberrations, self.pixelNumber, self.NA, self.wavelength, self.device, False, True)
        return phi

    def generateWavefrontError(self) -> torch.Tensor:
        """pupil.py:37-38."""
        return generateWavefrontError(self.aberrations, self.pixelNumber, self.NA, self.wavelength, self.device)


def throughFocusPupils(pixelNumber, wavelength, NA, aberrations, defocus_values, device, wavefront=False):
    """Stack of pupil functions that differ only in coefficient 4, complex64 [planes,pn,pn], written
    in place by ONE litho_pupil_stack launch per 64 planes.  The reference has no stack call; its counterpart is the loop
    `ab = aberrations.clone(); ab[4] = d; Pupil(pn, wavelength, NA, ab, device).generatePupilFunction()` per defocus
    value (pupil.py:88-111), which this reproduces bit for bit (the caller's `aberrations` is left alone, as the clones
    leave it).  wavefront=True: returns (W fp16 [planes,pn,pn], phi) -- every plane's fp16 wavefront error as well."""
    dev = nat.require_gpu(device)
    pn = int(pixelNumber)
    J = len(aberrations)
    if J < 5:
        raise IndexError(f"index 4 is out of bounds for dimension 0 with size {J}")
    planes = len(defocus_values)
    if planes < 1:
        raise ValueError("defocus_values must hold at least one plane")
    as_bits = lambda t: np.ascontiguousarray(t.detach().to(torch.float16).cpu().contiguous().view(torch.int16).numpy().astype(np.uint16))
    coeffs = as_bits(aberrations)
    defocus = as_bits(torch.as_tensor(defocus_values, dtype=torch.float64).to(torch.float16))   # `ab[4] = d` rounds d to fp16
    W = torch.empty((planes, pn, pn), dtype=torch.float16, device=dev) if wavefront else None
    phi = torch.empty((planes, pn, pn), dtype=torch.complex64, device=dev)
    with torch.cuda.device(dev):
        nat.check(nat.lib().litho_pupil_stack(coeffs.ctypes.data, J, defocus.ctypes.data, planes, pn, float(NA), float(wavelength),
                                              nat.ptr(W) if wavefront else None, nat.ptr(phi), nat.stream_ptr(dev)),
                  "litho_pupil_stack")
    return (W, phi) if wavefront else phi
