"""Effective shader clock of every kernel of a rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE run:
    python scripts/lab_clock.py <rocprofv3 output dir>
(GRBM_GUI_ACTIVE cycles / kernel duration; the second half of each kernel's dispatches, i.e. the warm ones)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
agg = collections.defaultdict(list)
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    name, t = dur[r["Dispatch_Id"]]
    agg[name.replace("void litho::", "").split("(")[0][:60]].append((float(r["Counter_Value"]), t))
for k, v in agg.items():
    v = v[len(v) // 2:]
    c = sum(x[0] for x in v) / len(v)
    t = sum(x[1] for x in v) / len(v)
    print(f"{k:60s} n={len(v):3d}  {t / 1e3:8.1f} us  GUI_ACTIVE {c:11.0f}  -> {c / t:.3f} GHz")
