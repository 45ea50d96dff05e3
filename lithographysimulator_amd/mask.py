"""Mask: same object API as the reference's mask.py, spectrum computed by HIP kernels."""
import torch

from . import _native as nat


class Mask:
    """Mirror of reference mask.py:3-90 (class Mask)."""

    def __init__(self, geometry: torch.Tensor = None, pixelSize: int = 25, device: torch.device = None):
        self.device = nat.pick_device(device, "mask")                      # mask.py:7-18
        if (geometry is None or type(geometry) is not torch.Tensor) or (
                len(geometry.size()) != 2 or geometry.size()[0] != geometry.size()[1]):
            # mask.py:20-27: never raises, falls back to the 64x64 four-bar demo
            print("Mask not defined or invalid. Check that it is a torch tensor and is square. Using demo instead.")
            self.pixelNumber = 64
            self.geometry = torch.zeros((64, 64), dtype=torch.int16, device=self.device)
            for c0 in (16, 25, 34, 43):
                self.geometry[9:55, c0:c0 + 4] = 1
        else:
            self.geometry = geometry.to(dtype=torch.int16, device=self.device)
            self.pixelNumber = self.geometry.size()[0]
        self.pixelSize = pixelSize
        self._pixelBound = self.pixelNumber / 2 * self.pixelSize
        self.deltaK = 4 / self.pixelNumber                                  # mask.py:34
        self._Kbound = self.pixelNumber / 2 * self.deltaK

    def fraunhofer(self, wavelength, fft: bool) -> torch.Tensor:
        """mask.py:37-40.  Only the FFT formulation is built (the O(pn^4) direct integral of
        mask.py:41-61 is outside the hot path)."""
        if not fft:
            raise NotImplementedError("the direct (non-FFT) Fraunhofer integral is not part of the MI355X engine; "
                                      "call fraunhofer(wavelength, True)")
        epsilon, N = self.calculateEpsilonN(self.deltaK, self.pixelSize, wavelength)
        return self._ffFraunhofer(epsilon, N)

    def _nearest2SqInt(self, input: float):
        """mask.py:63-65."""
        return nat.epsilon_n(1.0, 1.0, float(input))[1]

    def calculateEpsilonN(self, deltaK, pixelSize, wavelength):
        """mask.py:67-72; also callable unbound as Mask.calculateEpsilonN(self=mask, ...)."""
        return nat.epsilon_n(deltaK, pixelSize, wavelength)

    def _ffFraunhofer(self, epsilon, N: int) -> torch.Tensor:
        """mask.py:74-90 as one C-ABI call (bilinear scale, centred forward DFT, crop)."""
        dev = nat.require_gpu(self.device)
        pn = self.pixelNumber
        geo = self.geometry.contiguous()
        spec = torch.empty((pn, pn), dtype=torch.complex64, device=dev)
        ws = nat.workspace(dev, pn, N)
        with torch.cuda.device(dev):
            nat.check(nat.lib().litho_mask_spectrum(nat.ptr(geo), pn, float(epsilon), int(N), nat.ptr(spec),
                                                    nat.ptr(ws), ws.numel(), nat.stream_ptr(dev)),
                      "litho_mask_spectrum")
        return spec
