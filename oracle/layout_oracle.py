"""CPU restatement of the layout rasteriser (csrc/layout.hip) -- TEST INFRASTRUCTURE ONLY: tests/, smoke() and
bench.py's cpu_baseline leg may import this file; the product never does.

PARITY UNPINNED BY THE REFERENCE: quarterwave0/LithographySimulator has no layout import (README.md:20-22 names
GDSII import as a goal), so there is nothing of the reference's to check this against.  What pins it instead are the
closed-form cases of tests/test_layout_cpu.py (rectangles, a triangle, overlaps, edges through pixel centres): a pixel
is 1 when its centre (x0 + (c + 0.5) pixel, y0 + (r + 0.5) pixel) has a non-zero winding number with respect to the
counter-clockwise polygons; a centre exactly ON an edge counts as left of an edge's crossing only if strictly left
(half-open: left / bottom edges inside, right / top edges outside).  The arithmetic mirrors the kernel operation by
operation in float64 so that the comparison is bit for bit.
"""
import numpy as np


def rasterize_edges(edges: np.ndarray, pn: int, x0: float, y0: float, pixel: float) -> np.ndarray:
    """edges [n, 4] float64 (xa, ya, xb, yb) -> int16 [pn, pn]."""
    delta = np.zeros((pn, pn + 1), dtype=np.int64)
    rows = np.arange(pn, dtype=np.float64)
    yc = y0 + (rows + 0.5) * pixel
    for xa, ya, xb, yb in np.asarray(edges, dtype=np.float64).reshape(-1, 4):
        if not (np.isfinite([xa, ya, xb, yb]).all()) or ya == yb:
            continue
        d = 1 if yb > ya else -1
        ymin, ymax = min(ya, yb), max(ya, yb)
        hit = np.nonzero((ymin <= yc) & (yc < ymax))[0]
        if hit.size == 0:
            continue
        xc = xa + ((yc[hit] - ya) * (xb - xa)) / (yb - ya)
        k = np.clip(np.ceil((xc - x0) / pixel - 0.5), 0, pn).astype(np.int64)
        np.add.at(delta, (hit, np.zeros_like(k)), d)
        np.add.at(delta, (hit, k), -d)
    w = np.cumsum(delta[:, :pn], axis=1)
    return (w != 0).astype(np.int16)


def point_in_polygons(polygons, x: float, y: float) -> bool:
    """Independent check for single points: non-zero SUMMED winding number over all polygons by the classical crossing
    count -- the kernel's definition (a clockwise lobe cancels an overlapping counter-clockwise polygon)."""
    total = 0
    for q in polygons:
        q = np.asarray(q, dtype=np.float64)
        w = 0
        for (xa, ya), (xb, yb) in zip(q, np.roll(q, -1, axis=0)):
            if ya <= y < yb and (xa + (y - ya) * (xb - xa) / (yb - ya)) > x:
                w += 1
            elif yb <= y < ya and (xa + (y - ya) * (xb - xa) / (yb - ya)) > x:
                w -= 1
        total += w
    return total != 0
