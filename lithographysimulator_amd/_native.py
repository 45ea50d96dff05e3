"""ctypes binding of liblitho_abbe.so (C ABI: include/litho_abbe.h).

The HIP library is the product: if it is missing, or no HIP device is visible, every
compute entry point raises -- there is no CPU fallback."""
import contextlib
import ctypes
import os
import threading
from ctypes import POINTER, c_char_p, c_double, c_int, c_int64, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LITHO_ABBE_LIB") or os.path.join(_HERE, "lib", "liblitho_abbe.so")

LITHO_OK, E_ARG, E_NSMALL, E_WORKSPACE, E_HIP, E_INDEX = 0, -1, -2, -3, -4, -5
_lib = None

_SIGNATURES = {
    "litho_version": (c_int, []),
    "litho_target_arch": (c_char_p, []),
    "litho_last_error": (c_char_p, []),
    "litho_epsilon_n": (c_int, [c_double, c_double, c_double, POINTER(c_double), POINTER(c_int)]),
    "litho_source_bitmap": (c_int, [c_int, c_double, c_double, c_int, c_double, c_double, c_int, c_double,
                                    c_void_p, c_void_p]),
    "litho_source_compact": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, POINTER(c_int64), c_void_p]),
    "litho_pupil": (c_int, [c_void_p, c_int, c_int, c_double, c_double, c_int, c_void_p, c_void_p, c_void_p]),
    "litho_pupil_stack": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_double, c_double, c_void_p, c_void_p, c_void_p]),
    "litho_pupil_phase": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "litho_abbe_workspace_bytes": (c_int, [c_int, c_int, POINTER(c_size_t)]),
    "litho_abbe_accumulate": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_void_p,
                                      c_void_p, c_size_t, c_void_p]),
    "litho_abbe_accumulate_counted": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_int,
                                              c_void_p, c_void_p, c_size_t, c_void_p, POINTER(c_int64)]),
    "litho_abbe_accumulate_planned": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_int,
                                              c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, POINTER(c_int64)]),
    "litho_abbe_accumulate_opts": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_int,
                                           c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, POINTER(c_int64)]),
    "litho_abbe_embedded_size": (c_int, [c_int, c_int, POINTER(c_int)]),
    "litho_abbe_plan_dry_run": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_void_p]),
    "litho_abbe_field": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "litho_postprocess_size": (c_int, [c_int, c_double, POINTER(c_int)]),
    "litho_postprocess": (c_int, [c_void_p, c_int, c_int, c_double, c_void_p, c_void_p]),
    "litho_postprocess_resist": (c_int, [c_void_p, c_int, c_int, c_double, c_double, c_double, c_void_p, c_void_p, c_void_p]),
    "litho_mask_spectrum": (c_int, [c_void_p, c_int, c_double, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "litho_rasterize_work_bytes": (c_size_t, [c_int]),
    "litho_rasterize_edges": (c_int, [c_void_p, c_int64, c_int, c_double, c_double, c_double, c_void_p, c_size_t, c_void_p, c_void_p]),
    "litho_abbe_last_plan": (c_int, [POINTER(c_int64)]),
    "litho_abbe_last_kernels": (c_int, [c_void_p, c_void_p, c_size_t]),
    "litho_abbe_set_profiling": (c_int, [c_int]),
    "litho_abbe_last_profile": (c_int, [POINTER(c_double)]),
}


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP extension first (`make` at the repository root or "
                "`python -c 'import __graft_entry__ as g; g.build()'`).  There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        arch = handle.litho_target_arch()
        if arch != b"gfx950" and os.environ.get("LITHO_ALLOW_DIAG") != "1":
            # "gfx950-diag" = a timing-diagnostic build (scripts/build_variants.sh) whose results are wrong
            raise RuntimeError(f"{LIB_PATH} reports target {arch!r}, not b'gfx950': refusing a diagnostic build "
                               "(set LITHO_ALLOW_DIAG=1 for timing scripts only)")
        _lib = handle
    return _lib


def exported_symbols():
    return list(_SIGNATURES)


def check(rc, what):
    if rc == LITHO_OK:
        return
    if rc == E_NSMALL:
        # the reference dies with a tensor-size RuntimeError in this situation
        raise RuntimeError(f"{what}: FFT size N is smaller than the mask pixelNumber; pixelSize is too "
                           "large for this wavelength (the reference fails here as well)")
    if rc == E_INDEX:
        raise IndexError("index 4 is out of bounds for dimension 0 with size 4")
    if rc == E_HIP:
        raise RuntimeError(f"{what}: HIP error: {lib().litho_last_error().decode()}")
    if rc == E_WORKSPACE:
        raise RuntimeError(f"{what}: workspace too small")
    raise ValueError(f"{what}: unsupported or invalid argument (pn must be even, 2..16384; N a power of two, "
                     "16..16384)")


def require_gpu(device):
    if not isinstance(device, torch.device):
        device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError(f"lithographysimulator_amd computes on an MI355X (HIP) device only; got device "
                           f"'{device}'.  There is no CPU fallback.")
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device is visible to PyTorch-ROCm; there is no CPU fallback")
    return device


def pick_device(device, what):
    """The reference's constructors accept anything and fall back mps > cuda > cpu with a
    notice (mask.py:7-18 etc.).  Here: a torch.device is kept as is; otherwise the HIP device."""
    if type(device) is torch.device:
        return device
    if torch.cuda.is_available():
        d = torch.device("cuda", torch.cuda.current_device())
        print(f"No device defined for {what}! Using {torch.cuda.get_device_name(d)}.")
        return d
    print(f"No device defined for {what}! No HIP device is visible; compute calls will fail.")
    return torch.device("cpu")


def stream_ptr(device):
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return c_void_p(t.data_ptr())


_workspaces = {}
_workspaces_lock = threading.Lock()
WORKSPACE_CACHE_BYTES = 8 << 30      # the cache keeps at most this much per device (one entry always stays)


def workspace(device, pn, N):
    """Scratch buffer for the Abbe / field / mask-spectrum calls, cached per (device, pn, N, calling thread, stream): 0.3 GiB at
    256^2 .. 1.3 GiB at 2048^2, 4.6 GiB from 4096^2 up (litho_abbe_workspace_bytes).

    Lifetime: the cache only DROPS ITS OWN REFERENCE when it evicts an entry (least recently used first, once a device's
    entries exceed WORKSPACE_CACHE_BYTES) -- a workspace that a PlanCache holds (and therefore every HIP graph captured
    through that PlanCache, whose kernels carry the raw pointer) stays allocated for as long as the PlanCache lives.
    Entries of other devices are never touched."""
    dev_index = device.index if device.index is not None else torch.cuda.current_device()
    # One workspace per (device, size) AND per calling thread and stream: a workspace is scratch of the call that runs in it, so
    # two Python threads (ctypes releases the GIL inside the library) or two streams must never share one.  The single-threaded,
    # single-stream caller -- the reference's usage -- sees exactly one entry per size, as before.
    key = (dev_index, pn, N, threading.get_ident(), int(torch.cuda.current_stream(device).cuda_stream))
    with _workspaces_lock:
        ws = _workspaces.pop(key, None)
        if ws is None:
            nbytes = c_size_t(0)
            check(lib().litho_abbe_workspace_bytes(pn, N, ctypes.byref(nbytes)), "litho_abbe_workspace_bytes")
            mine = [k for k in _workspaces if k[0] == dev_index]               # insertion order = least recently used first
            held = sum(_workspaces[k].numel() for k in mine)
            for k in mine:
                if held + nbytes.value <= WORKSPACE_CACHE_BYTES:
                    break
                held -= _workspaces.pop(k).numel()
            ws = torch.empty(nbytes.value, dtype=torch.uint8, device=device)
        _workspaces[key] = ws                                                  # (re)insert as most recently used
    # The engine launches on torch's CURRENT stream, which need not be the stream the block was allocated on; the caching
    # allocator orders reuse only against the allocation stream.  Recording the use keeps an evicted (or cleared) workspace
    # from being handed out again while Abbe kernels queued on this stream still read or write it (round-4 advice).
    ws.record_stream(torch.cuda.current_stream(device))
    return ws


def epsilon_n(deltaK, pixelSize, wavelength):
    eps, N = c_double(0), c_int(0)
    check(lib().litho_epsilon_n(float(deltaK), float(pixelSize), float(wavelength), ctypes.byref(eps),
                                ctypes.byref(N)), "litho_epsilon_n")
    return eps.value, N.value


class PlanRecord(ctypes.Structure):
    """litho_abbe_plan (include/litho_abbe.h)."""
    _fields_ = [("words", ctypes.c_int32 * 16), ("valid", ctypes.c_int32), ("pn", ctypes.c_int32),
                ("N", ctypes.c_int32), ("planes", ctypes.c_int32)]


class Options(ctypes.Structure):
    """litho_abbe_options (include/litho_abbe.h): per-call launch-planner options; -1 = not set."""
    _names = ("coarse", "batch", "groups", "xchunk", "tile", "plane_chunk", "w64", "rect", "w64_8192", "xsplit", "xrect",
              "w64x", "gcombine", "rowpairs", "force_generic", "force_general", "poison", "embed", "split", "coopdma")
    _fields_ = [("size", ctypes.c_int32)] + [(n, ctypes.c_int32) for n in _names]

    @classmethod
    def make(cls, mapping):
        o = cls()
        o.size = ctypes.sizeof(cls)
        for n in cls._names:
            setattr(o, n, -1)
        for k, v in (mapping or {}).items():
            if k not in cls._names:
                raise KeyError(f"unknown engine option {k!r}; known: {', '.join(cls._names)}")
            setattr(o, k, int(v))
        return o


class Region(ctypes.Structure):
    """litho_abbe_region: a byte range of the workspace."""
    _fields_ = [("offset", ctypes.c_int64), ("bytes", ctypes.c_int64)]

    @property
    def end(self):
        return self.offset + self.bytes


class DryPart(ctypes.Structure):
    """litho_abbe_dry_part (include/litho_abbe.h)."""
    _regions = ("plan", "twtab", "twtab2", "slab_region", "slab_used", "ic_used", "chat_used", "gam_used", "T_region", "T_used",
                "recon_T_used", "embed_M", "embed_P", "embed_O")
    _fields_ = ([(n, ctypes.c_int32) for n in ("present", "run_size", "general", "variant", "coarse", "natural_box", "wave_y", "xkind",
                                               "batch", "planes_in_flight", "groups", "slabs", "xchunk", "tile")]
                + [("source_points", ctypes.c_int64), ("t_item_bytes", ctypes.c_int64)] + [(n, Region) for n in _regions])


class DryRun(ctypes.Structure):
    """litho_abbe_dry_run (include/litho_abbe.h)."""
    _fields_ = ([(n, ctypes.c_int32) for n in ("size", "status", "run_size", "nowrap", "split", "reserved")]
                + [("workspace_bytes", ctypes.c_int64), ("list_a", Region), ("list_b", Region), ("split_counts", Region),
                   ("part", DryPart * 2)])


def plan_dry_run(pn, N, planes, plan_words, split_words=None, options=None, cus=256, workspace_bytes=0):
    """litho_abbe_plan_dry_run: what an Abbe call WOULD do (kernel families, batching, embedded evaluation, the split of a
    partly wrapping source list, every workspace region it uses) for the given read-back words -- no device involved; works
    without a GPU.  plan_words: the 14 plan words; split_words: the split's ten words (needed when the call would split)."""
    pw = (ctypes.c_int32 * 14)(*[int(v) for v in plan_words])
    sw = (ctypes.c_int32 * 10)(*[int(v) for v in split_words]) if split_words is not None else None
    opts = Options.make(options) if options is not None else None
    res = DryRun()
    res.size = ctypes.sizeof(DryRun)
    rc = lib().litho_abbe_plan_dry_run(int(pn), int(N), int(planes), ctypes.byref(pw), ctypes.byref(sw) if sw is not None else None,
                                       ctypes.byref(opts) if opts is not None else None, int(cus), int(workspace_bytes),
                                       ctypes.byref(res))
    check(rc, "litho_abbe_plan_dry_run")
    return res


_option_stack = threading.local()


@contextlib.contextmanager
def engineOptions(**kw):
    """`with engineOptions(coarse=2, batch=7): ...` -- launch-planner options for every Abbe call of THIS thread inside the
    block (nested blocks merge, inner wins), passed through the C ABI's litho_abbe_options; nothing touches os.environ.
    Names = the LITHO_ABBE_* variables of DESIGN.md section 6 in lower case."""
    Options.make(kw)                                       # validate the names now
    stack = getattr(_option_stack, "v", None)
    if stack is None:
        stack = _option_stack.v = []
    stack.append(kw)
    try:
        yield
    finally:
        stack.pop()


def current_options(extra=None):
    """Options record for a call: the enclosing engineOptions blocks merged with the call's own `options=` mapping; None
    when nothing is set (the library then reads the environment, as before)."""
    merged = {}
    for kw in getattr(_option_stack, "v", None) or ():
        merged.update(kw)
    if extra:
        merged.update(extra)
    return Options.make(merged) if merged else None


def last_plan():
    arr = (c_int64 * 16)()
    lib().litho_abbe_last_plan(arr)
    keys = ("general", "box_row0", "box_col0", "box_rows", "box_cols", "batch", "launches", "variant",
            "planes_in_flight", "groups_per_plane", "xchunk", "fused_xpass", "coarse_grid", "wave_ypass", "natural_box",
            "planned_from_record")
    return dict(zip(keys, list(arr)))


def rasterize_work_bytes(pn: int) -> int:
    return int(lib().litho_rasterize_work_bytes(int(pn)))


def last_kernels():
    """(x-pass kernel, y-pass kernel) of the last Abbe call's source-point loop, as rocprofv3 names them."""
    x, y = ctypes.create_string_buffer(96), ctypes.create_string_buffer(96)
    check(lib().litho_abbe_last_kernels(x, y, 96), "litho_abbe_last_kernels")
    return x.value.decode(), y.value.decode()


def set_profiling(on: bool):
    lib().litho_abbe_set_profiling(1 if on else 0)


def last_profile():
    arr = (c_double * 8)()
    lib().litho_abbe_last_profile(arr)
    return {"xpass_ms": arr[0], "xpass_launches": int(arr[1]), "xpass_points": int(arr[2]),
            "ypass_ms": arr[3], "ypass_launches": int(arr[4]), "ypass_points": int(arr[5]),
            "wave_ypass": bool(arr[6]), "planes_in_flight": int(arr[7]),
            "xpass_kernel": last_kernels()[0], "ypass_kernel": last_kernels()[1]}
