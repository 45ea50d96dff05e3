"""Copy the summaries of a profile round from gpurun_out/<round>/ (scratch) into profiles/ (tracked):
    python scripts/collect_profiles.py r03 [commit of the snapshot]
bench lines, the rocprofv3 kernel stats of the default command, the PMC summaries, and profiles/traffic.json
(merged from the per-workload traffic_*.json of scripts/pmc_traffic.sh)."""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", R)
dst = os.path.join(ROOT, "profiles")
pairs = [("bench.json", f"{R}_bench.json"), ("bench_under_rocprof.json", f"{R}_bench_under_rocprof.json"),
         ("stats/bench_kernel_stats.csv", f"{R}_bench_kernel_stats.csv"), ("bench_cfg1.json", f"{R}_bench_cfg1.json"),
         ("bench_cfg2.json", f"{R}_bench_cfg2.json"), ("bench_cfg4_shard0of8.json", f"{R}_bench_cfg4_shard0of8.json"),
         ("bench_cfg3_shard0of8.json", f"{R}_bench_cfg3_shard0of8.json"), ("bench_cfg5_shard0of8.json", f"{R}_bench_cfg5_shard0of8.json")]
pairs += [(os.path.basename(p), f"{R}_" + os.path.basename(p)) for p in glob.glob(os.path.join(src, "pmc_cfg*_summary.txt"))]
for a, b in pairs:
    pa = os.path.join(src, a)
    if os.path.exists(pa) and os.path.getsize(pa) > 0:
        shutil.copy(pa, os.path.join(dst, b))
        print("copied", b)
traffic = {}
for p in sorted(glob.glob(os.path.join(src, "traffic_cfg*.json"))):
    try:
        traffic.update(json.load(open(p)))
    except Exception as exc:
        print("skipped", p, exc)
if traffic:
    import subprocess
    try:
        # (optional second argument: the commit the gpurun snapshot was taken at, when later commits -- documents only -- exist)
        commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
        if subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "lithographysimulator_amd"], text=True).strip():
            commit += "+uncommitted"
    except Exception:
        commit = None
    for v in traffic.values():
        v["commit"] = commit          # the source tree the counters were captured from (this script runs right after the capture)
    json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    print("traffic.json:", {k: (round(v["xpass_bytes_per_item"] / 1e6, 2), round(v["ypass_bytes_per_item"] / 1e6, 2)) for k, v in traffic.items()})

# single-GPU step times of this round's default bench line -> the inputs of bench.py's N-GPU prediction (round-5 advice: they were
# literals that drift without notice).  cfg4's whole-list time = 8 x its measured shard (the default line runs shard 0 of 8).
bj = os.path.join(dst, f"{R}_bench.json")
if os.path.exists(bj):
    try:
        line = json.loads([ln for ln in open(bj) if ln.startswith("{")][-1])
        single = {"cfg3": line.get("median_ms_per_step") or line["ms_per_step"]}
        for e in line.get("extra_workloads", []):
            name = e.get("workload", "")
            for k in ("cfg1", "cfg2", "cfg4", "cfg5"):
                if name.startswith(f"BASELINE {k}:") and "ms_per_step" in e:
                    ms = e.get("median_ms_per_step") or e["ms_per_step"]
                    single[k] = ms * 8.0 if k == "cfg4" else ms
        json.dump({"source": f"profiles/{R}_bench.json: median_ms_per_step of the headline and of every extra workload (cfg4 = 8 x shard 0/8)",
                   "single_gpu_ms": single}, open(os.path.join(dst, "prediction_inputs.json"), "w"), indent=1)
        print("prediction_inputs.json:", {k: round(v, 2) for k, v in single.items()})
    except Exception as exc:
        print("prediction inputs skipped:", exc)
