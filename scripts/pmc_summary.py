"""Summarise rocprofv3 --pmc CSV output: per-kernel mean of each counter (and duration)."""
import csv, glob, sys, collections, os
root = sys.argv[1]
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-40:]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("==", f)
    for k, d in agg.items():
        if "pass" not in k: continue
        print("  ", k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)):
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"].split("(")[0][-40:]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    print("==", f)
    for k, v in agg.items():
        if "pass" in k: print("  ", k, "mean_us", round(sum(v) / len(v), 1), "n", len(v))
