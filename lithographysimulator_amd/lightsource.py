"""LightSource: same object API as the reference's lightsource.py, sampled by a HIP kernel
that reproduces the reference's fp16 sigma-grid arithmetic."""
import ctypes

import torch

from . import _native as nat


class LightSource:
    """Mirror of reference lightsource.py:3-73."""

    def __init__(self, sigmaIn=0, sigmaOut=0.6, pixelNumber: int = 64, NA=0.7, shiftX=0, shiftY=0,
                 device: torch.device = None):
        self.device = nat.pick_device(device, "light source")              # lightsource.py:7-18
        self.pixelNumber = pixelNumber
        self.NA = NA
        self.sigmaInner = sigmaIn
        self.sigmaOuter = sigmaOut
        self.shiftX = shiftX
        self.shiftY = shiftY

    def _bitmap(self, kind, count, rotation):
        dev = nat.require_gpu(self.device)
        pn = int(self.pixelNumber)
        out = torch.empty((pn, pn), dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            nat.check(nat.lib().litho_source_bitmap(kind, float(self.sigmaInner), float(self.sigmaOuter), pn,
                                                    float(self.shiftX), float(self.shiftY), int(count),
                                                    float(rotation), nat.ptr(out), nat.stream_ptr(dev)),
                      "litho_source_bitmap")
        return out

    def generateAnnular(self) -> torch.Tensor:
        """lightsource.py:34-50: int64 0/1 bitmap [pn,pn]."""
        return self._bitmap(0, 1, 0.0)

    def generateQuasar(self, count, rotation) -> torch.Tensor:
        """lightsource.py:52-73."""
        return self._bitmap(1, count, rotation)


def sourceShiftsAsync(lightsource: torch.Tensor, pixelNumber: int):
    """The same compaction without the host read-back: returns (shifts [pn*pn,2] int32 with only the first S rows
    written, count = 1-element int32 device tensor holding S).  For abbeIntensity(..., count=...)."""
    dev = nat.require_gpu(lightsource.device)
    pn = int(pixelNumber)
    if lightsource.dim() != 2 or lightsource.shape[0] != pn or lightsource.shape[1] != pn:
        raise ValueError(f"the source bitmap must be [{pn},{pn}] (the mask's pixelNumber); got {tuple(lightsource.shape)}")
    bm = lightsource.to(torch.int64).contiguous()
    shifts = torch.empty((pn * pn, 2), dtype=torch.int32, device=dev)
    scratch = torch.empty(pn + 1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        nat.check(nat.lib().litho_source_compact(nat.ptr(bm), pn, nat.ptr(shifts), pn * pn, nat.ptr(scratch), None,
                                                 nat.stream_ptr(dev)), "litho_source_compact")
    return shifts, scratch[pn:pn + 1]


def sourceShifts(lightsource: torch.Tensor, pixelNumber: int) -> torch.Tensor:
    """imageformation.py:59: (argwhere(lightsource) - pn//2).int(), row-major, as int32 [S,2]
    on the bitmap's device."""
    dev = nat.require_gpu(lightsource.device)
    pn = int(pixelNumber)
    if lightsource.dim() != 2 or lightsource.shape[0] != pn or lightsource.shape[1] != pn:
        # the reference silently mis-shifts when the source grid differs from the mask's
        raise ValueError(f"the source bitmap must be [{pn},{pn}] (the mask's pixelNumber); got {tuple(lightsource.shape)}")
    bm = lightsource.to(torch.int64).contiguous()
    shifts = torch.empty((pn * pn, 2), dtype=torch.int32, device=dev)
    scratch = torch.empty(pn + 1, dtype=torch.int32, device=dev)
    count = ctypes.c_int64(0)
    with torch.cuda.device(dev):
        nat.check(nat.lib().litho_source_compact(nat.ptr(bm), pn, nat.ptr(shifts), pn * pn, nat.ptr(scratch),
                                                 ctypes.byref(count), nat.stream_ptr(dev)), "litho_source_compact")
    return shifts[:count.value]
