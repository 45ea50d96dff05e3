"""Shared constants/helpers for the parity tests (mirror tests/golden/make_golden.py)."""
import ctypes
import hashlib
import math
import os
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WL, NA, PS = 193.0, 0.7, 25
DEMO_AB = [0, 0, 0.01, 0, 100, 0.01, 0, 0.01, 0.01, 0.01]

SOURCE_CASES = {
    "circ": dict(kind="annular", sin=0.0, sout=0.5),
    "annular": dict(kind="annular", sin=0.4, sout=0.8),
    "quasar": dict(kind="quasar", sin=0.4, sout=0.8),
    "annular_shift": dict(kind="annular", sin=0.4, sout=0.8, sx=0.25, sy=-0.5),
    "quasar3_oddshift": dict(kind="quasar", sin=0.3, sout=0.9, sx=0.2, sy=-0.1, count=3, rot=0.3),
}
PUPIL_CASES = {
    "ideal": None,
    "defocus_p100": [0, 0, 0, 0, 100],
    "defocus_m200": [0, 0, 0, 0, -200],
    "defocus_p30": [0, 0, 0, 0, 30],
    "demo": DEMO_AB,
    "short3": [0.1, 0.2, 0.05],
    "terms15": [0, 0, 0, 1, 3, 0, 0, 1, 0, 0, 0.02, 0.03, 0.01, 0.5, 0.2],
}
# Tolerances (measured noise floor of the reference itself: fp32 vs complex128
# of the same algorithm = 4e-7 rel-to-max at 64^2/S=184, 2.8e-6 at 256^2/S=3233).
TOL_FIELD = 5e-6          # max|dE| / max|E|          single-point field
TOL_IMAGE_MAX = 2e-5      # max|dI| / max I           image, S <= 4096
TOL_IMAGE_L2 = 5e-6       # ||dI||_2 / ||I||_2
TOL_PHI = 5e-7            # abs, pupil function


def f16(v):
    return torch.tensor(v, dtype=torch.float16)


def rel_max(a, b):
    a = torch.as_tensor(a).double() if not torch.is_complex(torch.as_tensor(a)) else torch.as_tensor(a).to(torch.complex128)
    b = torch.as_tensor(b).double() if not torch.is_complex(torch.as_tensor(b)) else torch.as_tensor(b).to(torch.complex128)
    return float((a - b).abs().max() / b.abs().max())


def rel_l2(a, b):
    a = torch.as_tensor(a).double(); b = torch.as_tensor(b).double()
    return float(torch.linalg.norm(a - b) / torch.linalg.norm(b))


def sha256_packed(bitmap: np.ndarray) -> np.ndarray:
    packed = np.packbits(bitmap.astype(np.uint8))
    return np.frombuffer(hashlib.sha256(packed.tobytes()).digest(), dtype=np.uint8)


def unpack_bitmap(blob: np.ndarray, pn: int) -> np.ndarray:
    bits = np.unpackbits(np.frombuffer(zlib.decompress(blob.tobytes()), dtype=np.uint8))
    return bits[:pn * pn].reshape(pn, pn)


def subsample_bitmap(bitmap: torch.Tensor, K: int) -> torch.Tensor:
    pts = torch.argwhere(bitmap)
    S = pts.shape[0]
    idx = (torch.arange(K) * S) // K
    out = torch.zeros_like(bitmap)
    out[pts[idx, 0], pts[idx, 1]] = 1
    return out


def crop_center(img, crop=128):
    pn = img.shape[0]
    c0 = pn // 2 - crop // 2
    return img[c0:c0 + crop, c0:c0 + crop]


def load_c_oracle():
    path = os.path.join(ROOT, "oracle", "libabbe_ref.so")
    if not os.path.exists(path):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(path)
    lib.oracle_field.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.oracle_abbe_accumulate.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return lib


def c_oracle_field(lib, pupil, maskFT, N, dy, dx):
    pn = maskFT.shape[0]
    p = np.ascontiguousarray(pupil.numpy()); m = np.ascontiguousarray(maskFT.numpy())
    out = np.zeros((pn, pn), dtype=np.complex128)
    rc = lib.oracle_field(p.ctypes.data, m.ctypes.data, pn, N, int(dy), int(dx), out.ctypes.data)
    assert rc == 0
    return torch.from_numpy(out)
