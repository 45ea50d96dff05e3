"""imageformation: abbeImage / calculateFFTAerial with the reference's signatures
(imageformation.py:32-77), computed by the HIP engine behind the C ABI."""
import ctypes

import torch

from . import _native as nat
from .lightsource import sourceShifts
from .mask import Mask          # the reference forgets this import at module level


def calculateAerial(pupil, maskFT, fraunhoferConstant, pixelNumber, pixelSize, device):
    """imageformation.py:3-30: the O(pn^4) direct integral is outside the engine's scope."""
    raise NotImplementedError("the direct (non-FFT) aerial-image integral is not part of the MI355X engine; "
                              "use fft=True")


def calculateFFTAerial(pf, maskFFFT, pixelNumber, N):
    """imageformation.py:32-45: complex64 [pn,pn] field of one (already rolled) pupil."""
    dev = nat.require_gpu(maskFFFT.device)
    pn = int(pixelNumber)
    pf = pf.to(device=dev, dtype=torch.complex64).contiguous()
    m = maskFFFT.to(torch.complex64).contiguous()
    out = torch.empty((pn, pn), dtype=torch.complex64, device=dev)
    rc = nat.lib().litho_abbe_workspace_bytes(pn, int(N), ctypes.byref(ctypes.c_size_t(0)))
    nat.check(rc, "calculateFFTAerial")
    ws = nat.workspace(dev, pn, int(N))
    with torch.cuda.device(dev):
        nat.check(nat.lib().litho_abbe_field(nat.ptr(pf), nat.ptr(m), pn, int(N), nat.ptr(out), nat.ptr(ws),
                                             ws.numel(), nat.stream_ptr(dev)), "litho_abbe_field")
    return out


def abbeIntensity(maskFT, pupilF, shifts, N, out=None):
    """The loop of abbeImage (imageformation.py:54-67) for an explicit (dy,dx) list:
    returns / accumulates into the raw fp32 intensity [planes?,pn,pn] BEFORE post-processing.
    pupilF may be [pn,pn] or a through-focus stack [planes,pn,pn]."""
    dev = nat.require_gpu(maskFT.device)
    pn = maskFT.size()[0]
    m = maskFT.to(torch.complex64).contiguous()
    p = pupilF.to(device=dev, dtype=torch.complex64).contiguous()
    stacked = p.dim() == 3
    planes = p.shape[0] if stacked else 1
    sh = shifts.to(device=dev, dtype=torch.int32).contiguous()
    if out is None:
        out = torch.zeros((planes, pn, pn) if stacked else (pn, pn), dtype=torch.float32, device=dev)
    rc = nat.lib().litho_abbe_workspace_bytes(pn, int(N), ctypes.byref(ctypes.c_size_t(0)))
    nat.check(rc, "abbeImage")
    ws = nat.workspace(dev, pn, int(N))
    with torch.cuda.device(dev):
        nat.check(nat.lib().litho_abbe_accumulate(nat.ptr(m), nat.ptr(p), planes, nat.ptr(sh), sh.shape[0], pn,
                                                  int(N), nat.ptr(out), nat.ptr(ws), ws.numel(),
                                                  nat.stream_ptr(dev)), "litho_abbe_accumulate")
    return out


def postProcess(raw, epsilon):
    """imageformation.py:69-77: |.| -> bilinear resample by 1/epsilon -> zero pad."""
    dev = nat.require_gpu(raw.device)
    stacked = raw.dim() == 3
    planes = raw.shape[0] if stacked else 1
    pn = raw.shape[-1]
    n_out = ctypes.c_int(0)
    nat.check(nat.lib().litho_postprocess_size(pn, float(epsilon), ctypes.byref(n_out)), "litho_postprocess_size")
    r = raw.to(torch.float32).contiguous()
    out = torch.empty((planes, n_out.value, n_out.value) if stacked else (n_out.value, n_out.value),
                      dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        nat.check(nat.lib().litho_postprocess(nat.ptr(r), planes, pn, float(epsilon), nat.ptr(out),
                                              nat.stream_ptr(dev)), "litho_postprocess")
    return out


def abbeImage(mask, maskFT: torch.Tensor, pupilF: torch.Tensor, lightsource: torch.Tensor, pixelSize: int,
              deltaK: float, wavelength, fft: bool, device: torch.device, group=None):
    """Drop-in for imageformation.py:47-77.

    `group`: optional torch.distributed process group.  When given (or when a default group
    is initialised and LITHO_SHARD_SOURCES=1), the source-point list is split into contiguous
    shards, one per rank, and the partial intensities are summed with ONE all-reduce (RCCL
    over xGMI on MI355X) before the linear post-process."""
    if not fft:
        raise NotImplementedError("only the FFT formulation (fft=True) is built; the direct integral "
                                  "(imageformation.py:3-30) is outside the hot path")
    epsilon, N = Mask.calculateEpsilonN(self=mask, deltaK=deltaK, pixelSize=pixelSize, wavelength=wavelength)
    pixelNumber = maskFT.size()[0]
    dev = nat.require_gpu(device)
    maskFT = maskFT.to(dev)
    shifts = sourceShifts(lightsource.to(dev), pixelNumber)                # imageformation.py:59
    from .distributed import resolve_group, shard_bounds
    group = resolve_group(group)
    if group is not None:
        import torch.distributed as dist
        lo, hi = shard_bounds(shifts.shape[0], dist.get_rank(group), dist.get_world_size(group))
        shifts = shifts[lo:hi]
    image = abbeIntensity(maskFT, pupilF.to(dev), shifts, N)               # imageformation.py:62-67
    if group is not None:
        import torch.distributed as dist
        dist.all_reduce(image, op=dist.ReduceOp.SUM, group=group)
    return postProcess(image, epsilon)                                      # imageformation.py:69-77
