"""Per-image kernel timeline from a rocprofv3 kernel trace of a script that runs the same image over and over
(scripts/cfg1_timeline.py): the kernels of ONE steady-state image in launch order with their mean duration and the mean idle gap
in front of each, the busy fraction, and the per-image totals.
    python3 scripts/trace_summary.py <..._kernel_trace.csv> <images to average over (taken from the end of the trace)>"""
import csv
import re
import sys
from collections import defaultdict

path, images = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"^void ", "", r["Kernel_Name"])
        name = re.sub(r"litho::", "", name)
        name = re.sub(r"\(.*$", "", name)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
# an image ends with the post-process kernel
ends = [i for i, r in enumerate(rows) if r[2].startswith("k_postprocess")]
if len(ends) < images + 1:
    sys.exit(f"only {len(ends)} images in the trace")
first = ends[-images - 1] + 1
per = [rows[ends[-images - 1 + k] + 1: ends[-images + k] + 1] for k in range(images)]
lens = {len(p) for p in per}
print(f"# {path}: {len(ends)} images in the trace, the last {images} averaged; kernels per image: {sorted(lens)}")
if len(lens) != 1:
    sys.exit("images differ in their kernel sequence")
L = lens.pop()
tot_busy = tot_span = 0.0
print(f"{'#':>3} {'kernel':70s} {'dur us':>8} {'gap before us':>14}")
prev_end = [rows[first - 1][1]] * images
agg = defaultdict(float)
for j in range(L):
    d = sum(p[j][1] - p[j][0] for p in per) / images / 1e3
    g = sum(per[k][j][0] - (per[k][j - 1][1] if j else rows[ends[-images - 1 + k]][1]) for k in range(images)) / images / 1e3
    print(f"{j:3d} {per[0][j][2][:70]:70s} {d:8.2f} {g:14.2f}")
    tot_busy += d
    tot_span += d + g
    agg[per[0][j][2].split('<')[0]] += d
print(f"# per image: {L} kernels, busy {tot_busy:.1f} us, span (post-process to post-process) {tot_span:.1f} us, busy fraction {tot_busy / tot_span:.2f}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"#   {k:40s} {v:8.1f} us")
