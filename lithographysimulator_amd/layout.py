"""Layout import: GDSII stream files -> polygons -> the binary mask raster that `Mask` takes.

SURVEY.md section 8(f) row 4: the reference lists "GDSII import" among its unbuilt goals (README.md:20-22) and has no
code for it, so there is NO PARITY TARGET for this module: it is the caller side of `Mask(geometry, pixelSize)`
(mask.py:5-30 takes a square 0/1 raster and nothing else).  What is pinned instead: the record grammar and the 8-byte
excess-64 real format against the published known values of a UNITS record (tests/test_layout_cpu.py), the transforms
against hand-computed placements, and the device rasteriser bit for bit against its CPU restatement in the test tree.

Host side (this file, numpy): stream-format reader and writer, hierarchy flattening (SREF / AREF with reflection,
magnification, rotation), PATH -> outline polygon, BOX -> rectangle.  Device side (csrc/layout.hip through the C ABI):
`rasterizeLayout` -- a pixel is 1 when its CENTRE lies inside the union of the polygons (non-zero winding, half-open
on edges: a centre exactly on a left or bottom edge is inside, on a right or top edge outside).

Coordinates: GDSII database units are integers; `GdsLibrary.user_unit_m` is metres per database unit.  The raster
calls take nanometres: x = columns, y = rows, row 0 at the BOTTOM of the window (y grows with the row index), pixel
(r, c) has its centre at (x0 + (c + 0.5) pixel, y0 + (r + 0.5) pixel).
"""
import math
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# record types of the stream format (the subset a mask layout uses; everything else is skipped record by record)
HEADER, BGNLIB, LIBNAME, UNITS, ENDLIB, BGNSTR, STRNAME, ENDSTR = 0x00, 0x01, 0x02, 0x03, 0x04, 0x05, 0x06, 0x07
BOUNDARY, PATH, SREF, AREF, TEXT, LAYER, DATATYPE, WIDTH, XY, ENDEL = 0x08, 0x09, 0x0A, 0x0B, 0x0C, 0x0D, 0x0E, 0x0F, 0x10, 0x11
SNAME, COLROW, NODE, STRANS, MAG, ANGLE, PATHTYPE, BOX, BOXTYPE, BGNEXTN, ENDEXTN = (
    0x12, 0x13, 0x15, 0x1A, 0x1B, 0x1C, 0x21, 0x2D, 0x2E, 0x30, 0x31)
DT_NONE, DT_BITS, DT_INT2, DT_INT4, DT_REAL8, DT_ASCII = 0, 1, 2, 3, 5, 6


def real8_decode(b: bytes) -> float:
    """8-byte GDSII real: sign bit, 7-bit excess-64 exponent of 16, 56-bit mantissa (value = m / 2^56 * 16^(e - 64))."""
    if len(b) != 8:
        raise ValueError("a GDSII real is 8 bytes")
    sign = -1.0 if b[0] & 0x80 else 1.0
    exponent = (b[0] & 0x7F) - 64
    mantissa = int.from_bytes(b[1:], "big")
    return sign * (mantissa / float(1 << 56)) * (16.0 ** exponent)


def real8_encode(v: float) -> bytes:
    """Nearest 56-bit-mantissa value to the double (exact arithmetic): a decoded real encodes back to the same bytes."""
    from fractions import Fraction
    if v == 0.0:
        return bytes(8)
    sign = 0x80 if v < 0 else 0
    q = abs(Fraction(float(v)))
    exponent = 64
    while q >= 1:
        q /= 16
        exponent += 1
    while q < Fraction(1, 16):
        q *= 16
        exponent -= 1
    mantissa = int(q * (1 << 56) + Fraction(1, 2))
    if mantissa >= 1 << 56:                                    # rounding carried into the next hex digit
        mantissa >>= 4
        exponent += 1
    if not 0 <= exponent <= 127:
        raise OverflowError("value outside the range of a GDSII real")
    return bytes([sign | exponent]) + mantissa.to_bytes(7, "big")


@dataclass
class Transform:
    """STRANS / MAG / ANGLE of a reference: p -> origin + R(angle) (mag * reflect_x(p))."""
    reflect: bool = False
    mag: float = 1.0
    angle_deg: float = 0.0

    def matrix(self) -> np.ndarray:
        a = math.radians(self.angle_deg)
        # exact for the multiples of 90 degrees every mask layout uses
        c, s = {0.0: (1.0, 0.0), 90.0: (0.0, 1.0), 180.0: (-1.0, 0.0), 270.0: (0.0, -1.0)}.get(
            self.angle_deg % 360.0, (math.cos(a), math.sin(a)))
        m = np.array([[c, -s], [s, c]]) * self.mag
        if self.reflect:
            m = m @ np.array([[1.0, 0.0], [0.0, -1.0]])
        return m


@dataclass
class GdsElement:
    kind: str                                  # "boundary" | "box" | "path" | "sref" | "aref"
    layer: int = 0
    datatype: int = 0
    xy: np.ndarray = None                      # int64 [n, 2], database units
    width: int = 0                             # path
    pathtype: int = 0
    bgnextn: int = 0
    endextn: int = 0
    sname: str = ""                            # references
    transform: Transform = field(default_factory=Transform)
    cols: int = 1
    rows: int = 1


@dataclass
class GdsStructure:
    name: str
    elements: List[GdsElement] = field(default_factory=list)


@dataclass
class GdsLibrary:
    name: str = "LIB"
    user_unit: float = 1e-3                    # database unit in user units
    user_unit_m: float = 1e-9                  # database unit in metres
    structures: Dict[str, GdsStructure] = field(default_factory=dict)

    def top_structures(self) -> List[str]:
        referenced = {e.sname for s in self.structures.values() for e in s.elements if e.kind in ("sref", "aref")}
        return [n for n in self.structures if n not in referenced]


def _records(data: bytes):
    pos, n = 0, len(data)
    while pos + 4 <= n:
        length, rtype, dtype = struct.unpack_from(">HBB", data, pos)
        if length == 0:                                        # zero padding after ENDLIB (tape blocks)
            break
        if length < 4 or pos + length > n:
            raise ValueError(f"malformed GDSII record at byte {pos} (length {length})")
        yield rtype, dtype, data[pos + 4:pos + length]
        pos += length


def _ints(payload: bytes, size: int) -> List[int]:
    fmt = ">%d%s" % (len(payload) // size, "h" if size == 2 else "i")
    return list(struct.unpack(fmt, payload))


def _ascii(payload: bytes) -> str:
    return payload.rstrip(b"\0").decode("ascii", errors="replace")


def readGDSII(source) -> GdsLibrary:
    """Parse a GDSII stream (a path, or the bytes themselves).  TEXT and NODE elements and properties are skipped."""
    data = source if isinstance(source, (bytes, bytearray)) else open(source, "rb").read()
    lib = GdsLibrary()
    cur: Optional[GdsStructure] = None
    el: Optional[GdsElement] = None
    skipping = False                                           # inside a TEXT / NODE element
    saw_header = False
    for rtype, dtype, payload in _records(bytes(data)):
        if rtype == HEADER:
            saw_header = True
        elif rtype == LIBNAME:
            lib.name = _ascii(payload)
        elif rtype == UNITS:
            lib.user_unit, lib.user_unit_m = real8_decode(payload[:8]), real8_decode(payload[8:16])
        elif rtype == BGNSTR:
            cur = GdsStructure("")
        elif rtype == STRNAME:
            cur.name = _ascii(payload)
            lib.structures[cur.name] = cur
        elif rtype == ENDSTR:
            cur = None
        elif rtype in (BOUNDARY, BOX, PATH, SREF, AREF):
            el = GdsElement({BOUNDARY: "boundary", BOX: "box", PATH: "path", SREF: "sref", AREF: "aref"}[rtype])
        elif rtype in (TEXT, NODE):
            skipping = True
        elif rtype == ENDEL:
            if el is not None and cur is not None:
                cur.elements.append(el)
            el, skipping = None, False
        elif rtype == ENDLIB:
            break
        elif skipping or el is None:
            continue
        elif rtype == LAYER:
            el.layer = _ints(payload, 2)[0]
        elif rtype in (DATATYPE, BOXTYPE):
            el.datatype = _ints(payload, 2)[0]
        elif rtype == WIDTH:
            el.width = _ints(payload, 4)[0]
        elif rtype == PATHTYPE:
            el.pathtype = _ints(payload, 2)[0]
        elif rtype == BGNEXTN:
            el.bgnextn = _ints(payload, 4)[0]
        elif rtype == ENDEXTN:
            el.endextn = _ints(payload, 4)[0]
        elif rtype == XY:
            el.xy = np.array(_ints(payload, 4), dtype=np.int64).reshape(-1, 2)
        elif rtype == SNAME:
            el.sname = _ascii(payload)
        elif rtype == COLROW:
            el.cols, el.rows = _ints(payload, 2)[:2]
        elif rtype == STRANS:
            el.transform.reflect = bool(payload[0] & 0x80)
        elif rtype == MAG:
            el.transform.mag = real8_decode(payload[:8])
        elif rtype == ANGLE:
            el.transform.angle_deg = real8_decode(payload[:8])
    if not saw_header:
        raise ValueError("not a GDSII stream: no HEADER record")
    return lib


def _rec(rtype: int, dtype: int, payload: bytes = b"") -> bytes:
    if len(payload) % 2:
        payload += b"\0"
    return struct.pack(">HBB", 4 + len(payload), rtype, dtype) + payload


def writeGDSII(lib: GdsLibrary, path: Optional[str] = None) -> bytes:
    """Serialise a library (the subset readGDSII understands); returns the bytes and writes them to `path` if given."""
    stamp = struct.pack(">12h", *([2026, 1, 1, 0, 0, 0] * 2))
    out = [_rec(HEADER, DT_INT2, struct.pack(">h", 600)), _rec(BGNLIB, DT_INT2, stamp),
           _rec(LIBNAME, DT_ASCII, lib.name.encode("ascii")),
           _rec(UNITS, DT_REAL8, real8_encode(lib.user_unit) + real8_encode(lib.user_unit_m))]
    for s in lib.structures.values():
        out += [_rec(BGNSTR, DT_INT2, stamp), _rec(STRNAME, DT_ASCII, s.name.encode("ascii"))]
        for e in s.elements:
            kind = {"boundary": BOUNDARY, "box": BOX, "path": PATH, "sref": SREF, "aref": AREF}[e.kind]
            out.append(_rec(kind, DT_NONE))
            if e.kind in ("sref", "aref"):
                out.append(_rec(SNAME, DT_ASCII, e.sname.encode("ascii")))
                t = e.transform
                if t.reflect or t.mag != 1.0 or t.angle_deg != 0.0:
                    out.append(_rec(STRANS, DT_BITS, struct.pack(">H", 0x8000 if t.reflect else 0)))
                    if t.mag != 1.0:
                        out.append(_rec(MAG, DT_REAL8, real8_encode(t.mag)))
                    if t.angle_deg != 0.0:
                        out.append(_rec(ANGLE, DT_REAL8, real8_encode(t.angle_deg)))
                if e.kind == "aref":
                    out.append(_rec(COLROW, DT_INT2, struct.pack(">2h", e.cols, e.rows)))
            else:
                out.append(_rec(LAYER, DT_INT2, struct.pack(">h", e.layer)))
                out.append(_rec(BOXTYPE if e.kind == "box" else DATATYPE, DT_INT2, struct.pack(">h", e.datatype)))
                if e.kind == "path":
                    if e.pathtype:
                        out.append(_rec(PATHTYPE, DT_INT2, struct.pack(">h", e.pathtype)))
                    out.append(_rec(WIDTH, DT_INT4, struct.pack(">i", e.width)))
                    if e.pathtype == 4:
                        out += [_rec(BGNEXTN, DT_INT4, struct.pack(">i", e.bgnextn)), _rec(ENDEXTN, DT_INT4, struct.pack(">i", e.endextn))]
            xy = np.asarray(e.xy, dtype=np.int64).reshape(-1)
            out.append(_rec(XY, DT_INT4, struct.pack(">%di" % xy.size, *[int(v) for v in xy])))
            out.append(_rec(ENDEL, DT_NONE))
        out.append(_rec(ENDSTR, DT_NONE))
    out.append(_rec(ENDLIB, DT_NONE))
    blob = b"".join(out)
    if path:
        with open(path, "wb") as fh:
            fh.write(blob)
    return blob


MITRE_LIMIT = 2.0          # mitre length / half-width beyond which a path joint is bevelled


def pathOutline(xy: np.ndarray, width: float, pathtype: int = 0, bgnextn: float = 0.0, endextn: float = 0.0) -> np.ndarray:
    """Outline polygon of a PATH: the centre line offset by width / 2 on both sides, mitred joints.  Path type 0:
    flush ends, 2: ends extended by width / 2, 4: by BGNEXTN / ENDEXTN, 1 (round ends) is drawn as type 2.  Joints sharper
    than MITRE_LIMIT allows are bevelled.  Only SIMPLE outlines rasterise as drawn: a path that crosses itself, or whose
    inner corner folds over on a segment shorter than its width, still unions by summed winding numbers."""
    p = np.asarray(xy, dtype=np.float64)
    keep = np.ones(len(p), dtype=bool)
    keep[1:] = np.any(p[1:] != p[:-1], axis=1)                 # repeated points have no direction
    p = p[keep]
    if len(p) < 2 or width <= 0:
        return np.zeros((0, 2))
    hw = width / 2.0
    d = p[1:] - p[:-1]
    d /= np.linalg.norm(d, axis=1)[:, None]
    ext0, ext1 = {0: (0.0, 0.0), 1: (hw, hw), 2: (hw, hw), 4: (float(bgnextn), float(endextn))}.get(pathtype, (0.0, 0.0))
    p = p.copy()
    p[0] -= d[0] * ext0
    p[-1] += d[-1] * ext1
    nrm = np.stack([-d[:, 1], d[:, 0]], axis=1)               # left normal of every segment
    left, right = [], []
    for i in range(len(p)):
        if i == 0:
            left.append(p[0] + nrm[0] * hw); right.append(p[0] - nrm[0] * hw)
            continue
        if i == len(p) - 1:
            left.append(p[-1] + nrm[-1] * hw); right.append(p[-1] - nrm[-1] * hw)
            continue
        # mitre: the point at distance hw from both segments' centre lines, b * hw / (1 + n0.n1), length hw / cos(half turn).
        # Beyond MITRE_LIMIT * hw (a turn sharper than ~120 degrees) the joint is BEVELLED like layout tools do: the two
        # segment offsets themselves, so an acute or nearly reversing joint cannot throw a spike of 1e4 half-widths (and
        # a self-intersecting outline whose negative lobes would cancel neighbouring polygons in the summed-winding raster).
        b = nrm[i - 1] + nrm[i]
        den = 1.0 + float(np.dot(nrm[i - 1], nrm[i]))
        if den > 1e-9 and float(np.linalg.norm(b)) / den <= MITRE_LIMIT:
            left.append(p[i] + b * (hw / den)); right.append(p[i] - b * (hw / den))
        else:
            turn_left = float(d[i - 1][0] * d[i][1] - d[i - 1][1] * d[i][0]) > 0.0
            # outer side of the turn gets the two bevel corners; the inner side keeps the (clamped) mitre point
            inner = b * (hw / den) if den > 1e-9 else nrm[i] * 0.0
            lim = MITRE_LIMIT * hw
            ln = float(np.linalg.norm(inner))
            if ln > lim:
                inner = inner * (lim / ln)
            if turn_left:                                      # outer side = right
                left.append(p[i] + inner)
                right.append(p[i] - nrm[i - 1] * hw); right.append(p[i] - nrm[i] * hw)
            else:                                              # outer side = left
                left.append(p[i] + nrm[i - 1] * hw); left.append(p[i] + nrm[i] * hw)
                right.append(p[i] - inner)
    return np.array(left + right[::-1])


def flattenLayout(lib: GdsLibrary, top: Optional[str] = None, layers: Optional[Sequence[Tuple[int, int]]] = None,
                  max_depth: int = 64) -> List[np.ndarray]:
    """All polygons of structure `top` (default: the only top-level structure) with every reference expanded, in
    NANOMETRES (float64 [n, 2]), counter-clockwise.  `layers`: (layer, datatype) pairs to keep (datatype None = any)."""
    if top is None:
        tops = lib.top_structures()
        if len(tops) != 1:
            raise ValueError(f"the library has {len(tops)} top-level structures {tops}: name one")
        top = tops[0]
    if top not in lib.structures:
        raise KeyError(f"no structure {top!r} in the library")
    nm = lib.user_unit_m * 1e9

    def wanted(e: GdsElement) -> bool:
        return layers is None or any(e.layer == l and (d is None or e.datatype == d) for l, d in layers)

    out: List[np.ndarray] = []

    def walk(name: str, m: np.ndarray, o: np.ndarray, depth: int):
        if depth > max_depth:
            raise RecursionError(f"reference depth above {max_depth} (a cycle through {name!r}?)")
        for e in lib.structures[name].elements:
            if e.kind in ("boundary", "box"):
                if wanted(e) and e.xy is not None and len(e.xy) >= 3:
                    pts = e.xy.astype(np.float64)
                    if np.array_equal(pts[0], pts[-1]):
                        pts = pts[:-1]                          # the stream repeats the first vertex
                    out.append(pts @ m.T + o)
            elif e.kind == "path":
                if wanted(e) and e.xy is not None:
                    poly = pathOutline(e.xy, abs(e.width), e.pathtype, e.bgnextn, e.endextn)
                    if len(poly) >= 3:
                        out.append(poly @ m.T + o)
            elif e.kind in ("sref", "aref"):
                if e.sname not in lib.structures:
                    raise KeyError(f"structure {name!r} references the missing structure {e.sname!r}")
                t = e.transform.matrix()
                p = e.xy.astype(np.float64)
                if e.kind == "sref":
                    walk(e.sname, m @ t, p[0] @ m.T + o, depth + 1)
                else:
                    # the referenced cell is flattened ONCE (at the array's origin) and its polygons are translated to
                    # the cols x rows lattice points: a 1000 x 1000 array costs one recursion, not a million
                    dc, dr = (p[1] - p[0]) / max(e.cols, 1), (p[2] - p[0]) / max(e.rows, 1)
                    first = len(out)
                    walk(e.sname, m @ t, p[0] @ m.T + o, depth + 1)
                    cell = out[first:]
                    del out[first:]
                    ii, jj = np.meshgrid(np.arange(e.cols), np.arange(e.rows), indexing="xy")
                    offs = (ii.reshape(-1, 1) * dc + jj.reshape(-1, 1) * dr) @ m.T          # [cols * rows, 2], row-major in j
                    for q in cell:
                        out.extend(list(q[None, :, :] + offs[:, None, :]))

    walk(top, np.eye(2), np.zeros(2), 0)
    return [q for batch in _oriented_batches(out, nm) for q in batch]


def _oriented_batches(polygons, scale=1.0):
    """The polygons made counter-clockwise (a reflection turns them clockwise) and scaled, as [n, k, 2] arrays grouped
    by vertex count k (in order of first appearance): layouts are millions of rectangles, not a Python loop over them."""
    groups: Dict[int, List[np.ndarray]] = {}
    for q in polygons:
        q = np.asarray(q, dtype=np.float64).reshape(-1, 2)
        if len(q) >= 3:
            groups.setdefault(len(q), []).append(q)
    batches = []
    for k, items in groups.items():
        a = np.stack(items) * scale
        area2 = np.sum(a[:, :, 0] * np.roll(a[:, :, 1], -1, axis=1) - np.roll(a[:, :, 0], -1, axis=1) * a[:, :, 1], axis=1)
        cw = area2 < 0
        a[cw] = a[cw, ::-1]
        batches.append(a)
    return batches


def polygonEdges(polygons: Sequence[np.ndarray]) -> np.ndarray:
    """[n_edges, 4] float64 (x0, y0, x1, y1) of the closed polygons, every polygon made counter-clockwise first (so that
    overlapping polygons add winding numbers of the same sign: the raster is their UNION)."""
    rows = [np.concatenate([a, np.roll(a, -1, axis=1)], axis=2).reshape(-1, 4) for a in _oriented_batches(polygons)]
    return np.ascontiguousarray(np.concatenate(rows, axis=0)) if rows else np.zeros((0, 4))


def rasterizeLayout(polygons: Sequence[np.ndarray], pixelNumber: int, pixelSize: float, origin=None, device=None):
    """Binary mask raster (torch int16 [pn, pn] on `device`, what `Mask` takes) of polygons given in nanometres.
    origin = (x0, y0) of the window's lower-left corner; None centres the window on the polygons' bounding box."""
    import torch

    from . import _native as nat
    dev = nat.require_gpu(nat.pick_device(device, "layout"))
    pn = int(pixelNumber)
    edges = polygonEdges(polygons)
    if origin is None:
        if len(edges):
            lo, hi = edges[:, :2].min(axis=0), edges[:, :2].max(axis=0)
            ctr = (lo + hi) / 2.0
        else:
            ctr = np.zeros(2)
        origin = (float(ctr[0]) - pn * pixelSize / 2.0, float(ctr[1]) - pn * pixelSize / 2.0)
    geo = torch.empty((pn, pn), dtype=torch.int16, device=dev)
    work = torch.empty((nat.rasterize_work_bytes(pn),), dtype=torch.uint8, device=dev)
    ed = torch.from_numpy(edges.reshape(-1)).to(dev) if len(edges) else torch.zeros(4, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        nat.check(nat.lib().litho_rasterize_edges(nat.ptr(ed), int(len(edges)), pn, float(origin[0]), float(origin[1]),
                                                  float(pixelSize), nat.ptr(work), work.numel(), nat.ptr(geo),
                                                  nat.stream_ptr(dev)), "litho_rasterize_edges")
    return geo


def maskFromGDSII(source, pixelNumber: int, pixelSize: float, top: Optional[str] = None,
                  layers: Optional[Sequence[Tuple[int, int]]] = None, origin=None, device=None):
    """GDSII file -> `Mask` (the object abbeImage takes): read, flatten `top`, rasterise the chosen layers."""
    from .mask import Mask
    lib = source if isinstance(source, GdsLibrary) else readGDSII(source)
    geo = rasterizeLayout(flattenLayout(lib, top, layers), pixelNumber, pixelSize, origin, device)
    return Mask(geo, pixelSize, geo.device)
