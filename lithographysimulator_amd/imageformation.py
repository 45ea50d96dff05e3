"""imageformation: abbeImage / calculateFFTAerial with the reference's signatures
(imageformation.py:32-77), computed by the HIP engine behind the C ABI."""
import ctypes

import torch

from . import _native as nat
from .lightsource import sourceShifts, sourceShiftsAsync
from .mask import Mask          # the reference forgets this import at module level


class ShapeError(ValueError, RuntimeError):
    """Operand shapes that do not fit together.  The reference dies with a broadcasting RuntimeError at
    `pf * maskFFFT` (imageformation.py:34) in these cases; a raw device pointer must never see them."""


def _square(t, what):
    if t.dim() != 2 or t.shape[0] != t.shape[1]:
        raise ShapeError(f"{what} must be a square 2-D tensor; got {tuple(t.shape)}")
    return int(t.shape[0])


def calculateAerial(pupil, maskFT, fraunhoferConstant, pixelNumber, pixelSize, device):
    """imageformation.py:3-30: the O(pn^4) direct integral is outside the engine's scope."""
    raise NotImplementedError("the direct (non-FFT) aerial-image integral is not part of the MI355X engine; "
                              "use fft=True")


def calculateFFTAerial(pf, maskFFFT, pixelNumber, N):
    """imageformation.py:32-45: complex64 [pn,pn] field of one (already rolled) pupil."""
    pn = int(pixelNumber)
    if _square(maskFFFT, "maskFFFT") != pn or tuple(pf.shape) != (pn, pn):
        raise ShapeError(f"calculateFFTAerial: pf {tuple(pf.shape)} and maskFFFT {tuple(maskFFFT.shape)} must both be "
                         f"[{pn},{pn}] (pixelNumber)")
    dev = nat.require_gpu(maskFFFT.device)
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


class PlanCache:
    """Caller-held plan for SEQUENCES of images that share the pupil (stack) and the source -- many masks through one
    optical setting.  Every abbeImage / abbeIntensity call otherwise compacts the source bitmap and reads 56 bytes back
    to plan (pupil support box, shift extents, count): one host wait per image.  With a PlanCache the first call does
    that and records it; later calls with the SAME cache issue no compaction, no planning launch and never wait for the
    stream, so images queue back to back.  A shifted, off-axis source (LightSource(shiftX, shiftY), lightsource.py:5), whose
    list the planning call splits into a non-wrapping and a wrapping part, is split again by every planned call from the
    record (no read-back): the non-wrapping points keep the fast path (litho_abbe_last_plan()["planned_from_record"] == 3).

    Contract: reuse a cache only while the pupil tensor(s) and the source bitmap are unchanged -- call invalidate() (or
    make a new one) after changing either.  As a safety net the cache remembers, host-side and without any device
    access, WHICH tensors it was made for (storage address, shape, torch's in-place version counter, device) and plans
    afresh when a call arrives with different ones or after an in-place write; a tensor that was freed and whose
    address and counter happen to come back identical cannot be told apart, hence the contract.

    The cache also HOLDS the engine workspace its calls run in: a HIP graph captured from a planned call carries raw
    pointers into that workspace, so keep the PlanCache alive for as long as the graph is replayed (the process-wide
    workspace cache may evict its own reference at any time; this one keeps the memory allocated)."""

    def __init__(self):
        self.record = nat.PlanRecord()
        self.shifts = None          # compacted (dy,dx) list of the source bitmap (abbeImage)
        self.count = None           # its device-side count, until the first call has brought it to the host
        self.S = None
        self.workspace = None       # engine scratch of the planned calls (kept alive for captured graphs)
        self.identity = None        # what the record was made for (see _identity): abbeIntensity's pupil + shift list
        self.image_identity = None  # ... and abbeImage's pupil + source bitmap

    def invalidate(self):
        self.record.valid = 0
        self.shifts = self.count = self.S = self.identity = self.image_identity = None

    @property
    def valid(self):
        return bool(self.record.valid)


def _identity(*tensors):
    """Host-only fingerprint of the caller's tensors: (address, shape, in-place version, device) each."""
    return tuple(None if t is None else (t.data_ptr(), tuple(t.shape), t._version, str(t.device), t.dtype) for t in tensors)


def embeddedSize(pn: int, N: int) -> int:
    """Grid size the engine RUNS a pn x pn problem at (litho_abbe_embedded_size).  The specialised kernels (and the coarse grid)
    exist for pn = N and pn = N / 2; any other even size -- a 1000^2 or 3000^2 mask; 10 nm pixels, where N = 4 pn -- is
    evaluated embedded, inside the library: mask spectrum and pupil centred in a zero-padded N / 2 or N grid, the same shift
    list, the centre of the accumulated intensity added to the caller's image.  Identical sum, 1.7-3.6x faster than the generic
    kernels such sizes used to run on; options {"embed": 0} turns it off."""
    size = ctypes.c_int(0)
    nat.check(nat.lib().litho_abbe_embedded_size(int(pn), int(N), ctypes.byref(size)), "litho_abbe_embedded_size")
    return size.value


def _plan_workspace(plan, dev, pn, N, pupilF, shifts):
    """Engine scratch for a call at grid size pn; with a PlanCache, ITS workspace (graphs captured from planned calls hold raw
    pointers into it) and the identity check of the tensors the plan was made for."""
    if plan is None:
        return nat.workspace(dev, pn, N)
    if (plan.workspace is None or plan.workspace.device != dev or getattr(plan, "_ws_key", None) != (pn, N)):
        plan.workspace, plan._ws_key = nat.workspace(dev, pn, N), (pn, N)
    ident = _identity(pupilF) + ((shifts.data_ptr(), shifts._version, str(shifts.device)),)   # not the list's length: a
    # caller may pass the compacted list at its capacity first (with `count`) and as a [:S] view afterwards
    if plan.valid and plan.identity is not None and plan.identity != ident:
        plan.record.valid = 0                          # another pupil / source list, or an in-place write: plan afresh
    plan.identity = ident
    return plan.workspace


def abbeIntensity(maskFT, pupilF, shifts, N, out=None, count=None, plan=None, options=None):
    """The loop of abbeImage (imageformation.py:54-67) for an explicit (dy,dx) list:
    returns / accumulates into the raw fp32 intensity [planes?,pn,pn] BEFORE post-processing.
    pupilF may be [pn,pn] or a through-focus stack [planes,pn,pn].
    `count`: optional 1-element int32 DEVICE tensor holding the number of valid rows of `shifts`
    (sourceShiftsAsync); the call then returns (intensity, S) and the whole image path waits for the
    stream once.  `plan`: optional PlanCache (see there); the call then returns (intensity, S) as well.
    `options`: optional mapping of launch-planner options for THIS call (litho_abbe_options: coarse, batch, groups,
    xchunk, tile, plane_chunk, ...; see _native.engineOptions), merged over the enclosing engineOptions blocks.
    Mask sizes other than N and N / 2 run embedded in the next such grid, inside the library (embeddedSize)."""
    pn = _square(maskFT, "maskFT")
    if pupilF.dim() not in (2, 3) or tuple(pupilF.shape[-2:]) != (pn, pn) or (pupilF.dim() == 3 and pupilF.shape[0] < 1):
        # e.g. a default Pupil() (pixelNumber 64) with a 256^2 mask: the reference fails at pf * maskFFFT
        raise ShapeError(f"pupilF must be [{pn},{pn}] or [planes,{pn},{pn}] to match maskFT; got {tuple(pupilF.shape)} "
                         "(build the Pupil with mask.pixelNumber)")
    if shifts.dim() != 2 or shifts.shape[1] != 2:
        raise ShapeError(f"shifts must be [S,2] (dy,dx) pairs; got {tuple(shifts.shape)}")
    dev = nat.require_gpu(maskFT.device)
    m = maskFT.to(torch.complex64).contiguous()
    p = pupilF.to(device=dev, dtype=torch.complex64).contiguous()
    stacked = p.dim() == 3
    planes = p.shape[0] if stacked else 1
    sh = shifts.to(device=dev, dtype=torch.int32).contiguous()
    want = (planes, pn, pn) if stacked else (pn, pn)
    if out is None:
        out = torch.zeros(want, dtype=torch.float32, device=dev)
    elif (out.dtype != torch.float32 or not out.is_contiguous() or out.device != m.device
          or tuple(out.shape) != want):
        raise ShapeError(f"out must be a contiguous float32 tensor of shape {want} on {m.device}; got "
                         f"{out.dtype} {tuple(out.shape)} on {out.device}, contiguous={out.is_contiguous()}")
    rc = nat.lib().litho_abbe_workspace_bytes(pn, int(N), ctypes.byref(ctypes.c_size_t(0)))
    nat.check(rc, "abbeImage")
    if count is not None and (count.dtype != torch.int32 or count.numel() != 1 or count.device != m.device):
        raise ShapeError("count must be a 1-element int32 tensor on the mask's device")
    ws = _plan_workspace(plan, dev, pn, int(N), pupilF, shifts)
    opts = nat.current_options(options)
    with torch.cuda.device(dev):
        if opts is not None:
            S = ctypes.c_int64(0)
            nat.check(nat.lib().litho_abbe_accumulate_opts(nat.ptr(m), nat.ptr(p), planes, nat.ptr(sh),
                                                           nat.ptr(count) if count is not None else None, sh.shape[0],
                                                           pn, int(N), nat.ptr(out), nat.ptr(ws), ws.numel(),
                                                           nat.stream_ptr(dev),
                                                           ctypes.byref(plan.record) if plan is not None else None,
                                                           ctypes.byref(opts), ctypes.byref(S)),
                      "litho_abbe_accumulate_opts")
            return (out, S.value) if (plan is not None or count is not None) else out
        if plan is not None:
            S = ctypes.c_int64(0)
            nat.check(nat.lib().litho_abbe_accumulate_planned(nat.ptr(m), nat.ptr(p), planes, nat.ptr(sh),
                                                              nat.ptr(count) if count is not None else None, sh.shape[0],
                                                              pn, int(N), nat.ptr(out), nat.ptr(ws), ws.numel(),
                                                              nat.stream_ptr(dev), ctypes.byref(plan.record), ctypes.byref(S)),
                      "litho_abbe_accumulate_planned")
            return out, S.value
        if count is not None:
            S = ctypes.c_int64(0)
            nat.check(nat.lib().litho_abbe_accumulate_counted(nat.ptr(m), nat.ptr(p), planes, nat.ptr(sh), nat.ptr(count),
                                                              sh.shape[0], pn, int(N), nat.ptr(out), nat.ptr(ws),
                                                              ws.numel(), nat.stream_ptr(dev), ctypes.byref(S)),
                      "litho_abbe_accumulate_counted")
            return out, S.value
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


def resistContour(raw, epsilon, threshold, dose=1.0, return_image=False):
    """Constant-threshold resist model on the post-processed grid, fused into the post-process pass (the reference
    lists photoresist response as an open goal, README.md:21; nothing to be compatible with): uint8 mask, 1 where
    dose * image >= threshold.  `raw` is the accumulated intensity [pn,pn] or [planes,pn,pn] (abbeIntensity);
    with return_image=True the fp32 aerial image of the same pass comes back too: (image, resist)."""
    dev = nat.require_gpu(raw.device)
    stacked = raw.dim() == 3
    planes = raw.shape[0] if stacked else 1
    pn = raw.shape[-1]
    n_out = ctypes.c_int(0)
    nat.check(nat.lib().litho_postprocess_size(pn, float(epsilon), ctypes.byref(n_out)), "litho_postprocess_size")
    r = raw.to(torch.float32).contiguous()
    shape = (planes, n_out.value, n_out.value) if stacked else (n_out.value, n_out.value)
    resist = torch.empty(shape, dtype=torch.uint8, device=dev)
    image = torch.empty(shape, dtype=torch.float32, device=dev) if return_image else None
    with torch.cuda.device(dev):
        nat.check(nat.lib().litho_postprocess_resist(nat.ptr(r), planes, pn, float(epsilon), float(dose), float(threshold),
                                                     nat.ptr(image) if return_image else None, nat.ptr(resist),
                                                     nat.stream_ptr(dev)), "litho_postprocess_resist")
    return (image, resist) if return_image else resist


def bossungCurves(raw, epsilon, threshold, doses, pixelSize, row=None, column=None, exposed=False):
    """Process-window table of a through-focus stack (the caller-side driver config 5 implies; the
    reference has no such function -- its counterpart would be a loop over Pupil + abbeImage + a hand measurement):
    critical dimension in nanometres [len(doses), planes] of the feature that crosses (row, column) of the
    post-processed grid (default: its centre), measured along that row on the resist contour dose * image >= threshold
    (one fused post-process + threshold pass per dose, every focal plane at once).  The feature is the run of UNEXPOSED
    pixels through the point (a line of a dark-field line/space pattern; exposed=True: the run of exposed pixels, a space
    or contact); 0 where the point is of the other kind.  `raw` = abbeIntensity of the stack, [planes, pn, pn]."""
    if raw.dim() != 3:
        raise ShapeError(f"bossungCurves takes the accumulated intensity of a stack [planes,pn,pn]; got {tuple(raw.shape)}")
    table = []
    for dose in doses:
        contour = resistContour(raw, epsilon, threshold, dose=float(dose))          # uint8 [planes, n, n]
        n = contour.shape[-1]
        r = n // 2 if row is None else int(row)
        c = n // 2 if column is None else int(column)
        line = contour[:, r, :].to(torch.int32)                                     # [planes, n]
        want = 1 if exposed else 0
        same = (line == want)
        # extent of the run of `want` pixels through column c: first differing pixel on either side
        idx = torch.arange(n, device=line.device).expand_as(line)
        left_stop = torch.where(~same & (idx < c), idx, torch.full_like(idx, -1)).max(dim=1).values
        right_stop = torch.where(~same & (idx > c), idx, torch.full_like(idx, n)).min(dim=1).values
        width = (right_stop - left_stop - 1).to(torch.float32) * float(pixelSize)
        table.append(torch.where(same[:, c], width, torch.zeros_like(width)))
    return torch.stack(table)


def _all_reduce_sum(image, group):
    """ONE collective per image/stack.  RCCL ("nccl" backend on ROCm) reduces the device tensor in
    place over xGMI; a gloo group (CPU tests, or several ranks sharing one GPU) goes through a host copy."""
    import torch.distributed as dist
    if dist.get_backend(group) == "gloo" and image.is_cuda:
        host = image.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        image.copy_(host)
    else:
        dist.all_reduce(image, op=dist.ReduceOp.SUM, group=group)
    return image


def abbeImage(mask, maskFT: torch.Tensor, pupilF: torch.Tensor, lightsource: torch.Tensor, pixelSize: int,
              deltaK: float, wavelength, fft: bool, device: torch.device, group=None, normalize: bool = False,
              plan_cache: PlanCache = None, options=None):
    """Drop-in for imageformation.py:47-77.  `pupilF` may also be a through-focus stack [planes,pn,pn] (BASELINE
    config 5; the reference's counterpart is a Python loop over Pupil(...) + abbeImage(...)), in which case the
    result is [planes,pn',pn'].

    `group`: optional torch.distributed process group.  When given (or when a default group
    is initialised and LITHO_SHARD_SOURCES=1), the source-point list is split into contiguous
    shards, one per rank, and the partial intensities are summed with ONE all-reduce (RCCL
    over xGMI on MI355X) before the linear post-process.

    `normalize`: divide by the number of source points S (the reference returns raw sums, Q7).

    `plan_cache`: optional PlanCache for sequences of images with the same pupil and source (single GPU): from the second
    call on, no source compaction, no planning launches and no host wait.  Not combinable with `group`.

    `options`: optional mapping of launch-planner options for this call (see abbeIntensity)."""
    if not fft:
        raise NotImplementedError("only the FFT formulation (fft=True) is built; the direct integral "
                                  "(imageformation.py:3-30) is outside the hot path")
    epsilon, N = Mask.calculateEpsilonN(self=mask, deltaK=deltaK, pixelSize=pixelSize, wavelength=wavelength)
    pixelNumber = _square(maskFT, "maskFT")                                 # imageformation.py:54
    if pupilF.dim() not in (2, 3) or tuple(pupilF.shape[-2:]) != (pixelNumber, pixelNumber):
        raise ShapeError(f"pupilF must be [{pixelNumber},{pixelNumber}] or [planes,{pixelNumber},{pixelNumber}] to "
                         f"match maskFT; got {tuple(pupilF.shape)} (build the Pupil with mask.pixelNumber)")
    if tuple(lightsource.shape) != (pixelNumber, pixelNumber):
        # the reference silently mis-shifts when the source grid differs from the mask's
        raise ShapeError(f"the source bitmap must be [{pixelNumber},{pixelNumber}] (the mask's pixelNumber); got "
                         f"{tuple(lightsource.shape)}")
    dev = nat.require_gpu(device)
    maskFT = maskFT.to(dev)
    from .distributed import resolve_group, shard_bounds
    group = resolve_group(group)
    if group is not None and plan_cache is not None:
        raise ValueError("plan_cache is for single-GPU image sequences; a sharded call (group=...) plans per rank and per "
                         "call -- pass one or the other")
    if plan_cache is not None:
        planes = pupilF.shape[0] if pupilF.dim() == 3 else 1
        r = plan_cache.record
        ident = _identity(pupilF, lightsource)
        if (plan_cache.shifts is None or not plan_cache.valid or (r.pn, r.N, r.planes) != (pixelNumber, int(N), planes)
                or plan_cache.image_identity != ident):
            plan_cache.invalidate()
            plan_cache.image_identity = ident
            plan_cache.shifts, plan_cache.count = sourceShiftsAsync(lightsource.to(dev), pixelNumber)
            image, total = abbeIntensity(maskFT, pupilF.to(dev), plan_cache.shifts, N, count=plan_cache.count, plan=plan_cache,
                                         options=options)
            plan_cache.S, plan_cache.count = total, None           # the count is on the host now (and in the record)
        else:
            image, total = abbeIntensity(maskFT, pupilF.to(dev), plan_cache.shifts[:plan_cache.S], N, plan=plan_cache,
                                         options=options)
    elif group is None:
        # single GPU: the source count never visits the host on its own -- one stream wait per image
        shifts, count = sourceShiftsAsync(lightsource.to(dev), pixelNumber)          # imageformation.py:59
        image, total = abbeIntensity(maskFT, pupilF.to(dev), shifts, N, count=count, options=options)  # imageformation.py:62-67
    else:
        import torch.distributed as dist
        shifts = sourceShifts(lightsource.to(dev), pixelNumber)            # imageformation.py:59
        total = shifts.shape[0]
        lo, hi = shard_bounds(total, dist.get_rank(group), dist.get_world_size(group))
        image = abbeIntensity(maskFT, pupilF.to(dev), shifts[lo:hi], N, options=options)    # imageformation.py:62-67
        _all_reduce_sum(image, group)
    if normalize and total > 0:
        image /= float(total)
    return postProcess(image, epsilon)                                      # imageformation.py:69-77
