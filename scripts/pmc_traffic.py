"""Fold the FETCH_SIZE / WRITE_SIZE passes of scripts/pmc_traffic.sh into bytes per T item (source point x plane):
    python scripts/pmc_traffic.py <pmc dir> <bench json of the same command> <workload>

Memory-side bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports both counters in KiB, and on gfx950
FETCH_SIZE tallies the 128-byte requests of wide streaming reads at 64 bytes (MI355X_MICROARCH.md, "HBM"), so it is
doubled; WRITE_SIZE is taken as reported.  The factor-2 correction is an UPPER bound for the 8-byte-per-lane T and M
loads of these kernels (the guide calibrates it on 16-byte loads only)."""
import collections
import csv
import glob
import json
import os
import sys

GEOMETRY_KEYS = ("batch", "groups_per_plane", "xchunk", "planes_in_flight", "coarse_grid", "variant")   # = bench.py's
root, bench_json, workload = sys.argv[1:4]
bench = json.loads([ln for ln in open(bench_json) if ln.startswith("{")][-1])
cfg = bench["config"]
# every execution of the step in that process: timed step + profiled step (+ warm-up and the event pass of short steps, if any)
items_total = cfg.get("step_executions", 2) * cfg["source_points"] * cfg["planes"]


def per_class(counter, sub):
    tot = collections.defaultdict(float)
    n = collections.defaultdict(int)
    files = glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            # per-source-point kernels only (the once-per-image reconstruction transforms are not T-item work)
            cls = ("xpass" if any(k in name for k in ("k_xpass_abbe", "k_xpass_split", "k_xpass_rect", "k_xpass_w64"))
                   else "ypass" if any(k in name for k in ("k_ypass_wave", "k_ypass_rect", "k_ypass_pair",
                                                           "k_ypass_acc", "k_ypass_w64", "k_ypass_coop")) else None)
            if cls:
                tot[cls] += float(row["Counter_Value"])
                n[cls] += 1
    return tot, n


fetch, nf = per_class("FETCH_SIZE", "fetch")
write, nw = per_class("WRITE_SIZE", "write")
out = {"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `python3 bench.py --workload {workload} "
                 f"--steps 1 --warmup 0 --no-cpu-baseline --points {cfg['source_points']}` (consecutive source points, "
                 f"bench batching: {cfg['plan']['batch']} points x {cfg['plan']['planes_in_flight']} planes per launch pair)",
       "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes / T items (source point x plane); FETCH doubled per the "
                  "gfx950 note in MI355X_MICROARCH.md (upper bound for 8-byte loads)",
       "items": items_total,
       # launch geometry the bytes per item depend on (a host-side change of batch / groups / chunk changes them without
       # changing a kernel name; the y-pass kernel's name carries the tile width): bench.py drops the entry on any mismatch
       "geometry": {k: cfg["plan"][k] for k in GEOMETRY_KEYS}}
for cls in ("xpass", "ypass"):
    # the kernel the counters belong to, as the library names it in the same bench line (bench.py drops the entry when a
    # later build launches another kernel for this workload)
    out[cls + "_kernel"] = bench["roofline"]["kernels"][cls]["kernel"]
    out[cls + "_fetch_KiB_per_item"] = fetch[cls] / items_total
    out[cls + "_write_KiB_per_item"] = write[cls] / items_total
    out[cls + "_bytes_per_item"] = (2 * fetch[cls] + write[cls]) * 1024 / items_total
    out[cls + "_dispatches"] = nf[cls]
print(json.dumps({workload: out}, indent=1))
