"""Synthetic mask geometries: integer-only so that the build container,
the GPU box and the golden-vector generator all regenerate identical inputs."""
import numpy as np
import torch


def bernoulli_mask(pn: int, seed: int = 1234) -> torch.Tensor:
    """geometry[y,x] = bit 31 of ((x*73856093) xor (y*19349663) xor seed)*2654435761 mod 2^32."""
    x = np.arange(pn, dtype=np.uint64)[None, :]
    y = np.arange(pn, dtype=np.uint64)[:, None]
    m32 = np.uint64(0xFFFFFFFF)
    v = ((x * np.uint64(73856093)) & m32) ^ ((y * np.uint64(19349663)) & m32) ^ np.uint64(seed)
    v = (v * np.uint64(2654435761)) & m32
    return torch.from_numpy(((v >> np.uint64(31)) & np.uint64(1)).astype(np.int16))


def lines_mask(pn: int) -> torch.Tensor:
    """The reference's 64x64 four-bar demo pattern (mask.py:24-27) scaled by k = pn/64."""
    if pn % 64:
        raise ValueError("lines_mask needs pn to be a multiple of 64")
    k = pn // 64
    g = torch.zeros((pn, pn), dtype=torch.int16)
    for c0 in (16, 25, 34, 43):
        g[9 * k:55 * k, c0 * k:(c0 + 4) * k] = 1
    return g
