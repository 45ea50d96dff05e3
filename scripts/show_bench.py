"""Short summary of a bench.py JSON line: python scripts/show_bench.py file.json"""
import json
import sys

r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rl = r["roofline"]
print(f"value {r['value']:.4e} {r['unit']}  {r['ms_per_step']:.1f} ms/step  n_gpus {r['n_gpus']}")
print(f"roofline: {rl['kernel']}  frac {rl['frac']:.3f}  avg launch {rl['avg_launch_ms'] * 1e3:.1f} us  time share {rl.get('kernel_time_frac')}"
      f"  fabric_frac {rl.get('fabric_frac')}  eff40/peak {rl.get('effective_40B_over_peak'):.2f}  copy {rl.get('copy_ceiling_GBs')}")
for k, v in rl["kernels"].items():
    print(f"   {k}: {v['kernel']}  {v['avg_launch_ms'] * 1e3:.1f} us x {v['launches']}  ({v['items_per_launch']:.1f} items)")
if "cpu_baseline" in r:
    cb = r["cpu_baseline"]
    print(f"cpu_baseline {cb['value']:.3e} on {cb['cores']} threads, parity {cb['gpu_vs_cpu_rel_to_max']:.2e} on {cb.get('parity_path')}")
if "ranks" in r:
    print("ranks:", {k: v for k, v in r["ranks"].items() if k != "note"})
for e in r.get("extra_workloads", []):
    if "error" in e:
        print("  extra", e["workload"], "ERROR", e["error"])
        continue
    print(f"  extra {e['workload'][:40]:40s} {e['ms_per_step']:10.2f} ms/step  {e['value']:.3e}  {e['dominant_kernel']} "
          f"(time {e['dominant_kernel_time_frac']:.2f}, valu {e['dominant_kernel_valu_frac']:.2f})"
          + (f"  [sequence with PlanCache: {e['sequence_with_plan_cache']['ms_per_image']:.3f} ms, {e['sequence_with_plan_cache']['value']:.3e}]"
             if "sequence_with_plan_cache" in e else ""))
