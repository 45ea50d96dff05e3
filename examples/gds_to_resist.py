"""GDSII layout -> mask -> partially coherent aerial image -> constant-threshold resist contour, on one MI355X.

    python examples/gds_to_resist.py [layout.gds] [--top NAME] [--layer 7 --datatype 0] [--pn 512] [--pixel 25]

Without a file it writes a small line/space layout (a 12-bar grating placed by an AREF plus a wire drawn as a PATH) to
/tmp and uses that.  Prints the image statistics and the printed line width on the centre row; saves nothing."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lithographysimulator_amd as L                                     # noqa: E402
from lithographysimulator_amd import layout as LY                       # noqa: E402


def demo_layout(path):
    lib = LY.GdsLibrary("DEMO", 1e-3, 1e-9)                              # database unit = 1 nm
    bar = LY.GdsStructure("BAR")
    bar.elements.append(LY.GdsElement("boundary", layer=7, datatype=0,
                                      xy=np.array([[0, 0], [150, 0], [150, 6000], [0, 6000], [0, 0]])))
    top = LY.GdsStructure("TOP")
    top.elements.append(LY.GdsElement("aref", sname="BAR", cols=12, rows=1,
                                      xy=np.array([[3000, 3400], [3000 + 12 * 500, 3400], [3000, 3400 + 6000]])))
    top.elements.append(LY.GdsElement("path", layer=7, datatype=0, width=200, pathtype=2,
                                      xy=np.array([[3000, 2400], [9350, 2400]])))
    lib.structures["BAR"], lib.structures["TOP"] = bar, top
    LY.writeGDSII(lib, path)
    return path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("gds", nargs="?")
    ap.add_argument("--top")
    ap.add_argument("--layer", type=int, default=7)
    ap.add_argument("--datatype", type=int, default=0)
    ap.add_argument("--pn", type=int, default=512)
    ap.add_argument("--pixel", type=float, default=25.0)
    ap.add_argument("--threshold", type=float, default=0.3, help="resist threshold as a fraction of the clear-field intensity")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    wl, na = 193.0, 0.7
    path = a.gds or demo_layout("/tmp/litho_demo.gds")
    mask = L.maskFromGDSII(path, a.pn, a.pixel, top=a.top, layers=[(a.layer, a.datatype)], device=dev)
    print(f"{path}: {int(mask.geometry.sum())} of {a.pn * a.pn} mask pixels set")
    source = L.LightSource(0.4, 0.8, a.pn, na, device=dev).generateAnnular()
    pupil = L.Pupil(a.pn, wl, na, torch.tensor([0, 0, 0, 0, 30], dtype=torch.float16), dev).generatePupilFunction()
    mft = mask.fraunhofer(wl, True)
    eps, N = mask.calculateEpsilonN(mask.deltaK, a.pixel, wl)
    shifts = L.sourceShifts(source, a.pn)
    raw = L.abbeIntensity(mft, pupil, shifts, N)                          # the Abbe sum (the hot path)
    # clear-field level: the same optics over an all-open mask
    clear = L.abbeIntensity(L.Mask(torch.ones((a.pn, a.pn), dtype=torch.int16), a.pixel, dev).fraunhofer(wl, True), pupil, shifts, N)
    level = float(clear[a.pn // 2, a.pn // 2])
    image, contour = L.resistContour(raw, eps, a.threshold * level, return_image=True)
    image = image / level
    print(f"{shifts.shape[0]} source points, FFT size {N}, image {tuple(image.shape)}: min {float(image.min()):.3f} max {float(image.max()):.3f} (clear field = 1)")
    row = contour[contour.shape[0] // 2].cpu().numpy().astype(np.int8)
    edges = np.flatnonzero(np.diff(row))
    if len(edges) >= 2:
        widths = (edges[1::2] - edges[0::2][:len(edges[1::2])]) * a.pixel
        print(f"centre row: {len(widths)} exposed runs above threshold, widths (nm): {widths[:12].tolist()}")
    print(f"resist contour: {int(contour.sum())} of {contour.numel()} pixels above {a.threshold:.2f} x clear field")


if __name__ == "__main__":
    main()
