#!/usr/bin/env python3
"""Headline benchmark: Abbe source-points x image-pixels per second on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg1..cfg5] [--shard i/n] [--points K]

One "step" = one complete abbeImage call (source-list compaction, Abbe accumulation over every source point --
and every through-focus plane -- of the configuration, all-reduce when N > 1, post-process) on synthetic inputs
that are resident in HBM when the timed region starts.  Default workload = BASELINE config 3, the one the
roofline target is quoted on: 2048x2048 bernoulli mask, quasar source sigma 0.4-0.8 (S = 198,108), 10-term
Zernike-aberrated pupil.  With N > 1 (one rank per GPU, RCCL) the source list is split into N contiguous shards
and the partial intensities are summed by ONE all-reduce: total work is fixed -> "strong".

Launching: under torch.distributed.run the ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the
environment.  Started as a plain process with --gpus N > 1, this file starts the N ranks ITSELF as fresh child
processes (before anything in the parent touches the GPU), waits for them and relays rank 0's JSON line.

Rank 0 prints one JSON line (contract in the task statement) carrying
  roofline     the dominant kernel against the bound the counters show: VALU issue.  achieved = nominal FFT flops
               (5 N log2 N per transformed line) of one launch / its HIP-event-timed duration, peak = 157.3
               TFLOP/s fp32 vector.  The HBM view sits beside it: `traffic` = PMC-measured memory-side bytes per
               launch of THIS workload (profiles/traffic.json, made by scripts/pmc_traffic.sh from rocprofv3 --pmc
               passes over this very command), `hbm.measured_frac` = traffic / time / 8 TB/s, and
               `hbm.effective_40B` = the SURVEY 8d 40-byte model over kernel time, labelled effective because
               most of those bytes are never moved (pupil-box pruning, on-chip accumulators, cache-resident M/P);
  cpu_baseline (N = 1) the oracle's torch-CPU op chain, i.e. a port of the reference loop, on a bounded sample.
"""
import argparse
import json
import math
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WL, NA, PS = 193.0, 0.7, 25
DEMO_AB = [0, 0, 0.01, 0, 100, 0.01, 0, 0.01, 0.01, 0.01]
DEFOCUS_NM = [-310 + 20 * k for k in range(32)]           # config 5
WORKLOADS = {
    # name: (pn, source kind, aberrations, planes, description)
    "cfg1": (256, "circ", None, 1, "256x256 bernoulli mask, circular source sigma 0.5, ideal pupil (the reference's CPU-runnable case)"),
    "cfg2": (1024, "annular", [0, 0, 0, 0, 100], 1, "1024x1024 bernoulli mask, annular 0.4-0.8, defocus-only pupil"),
    "cfg3": (2048, "quasar", DEMO_AB, 1, "2048x2048 bernoulli mask, quasar(4,-pi/8) 0.4-0.8, 10-term Zernike pupil"),
    "cfg4": (4096, "annular", [0, 0, 0, 0, 100], 1, "4096x4096 bernoulli mask, annular 0.4-0.8, defocus-only pupil"),
    "cfg5": (2048, "quasar", DEMO_AB, 32, "2048x2048 bernoulli mask x 32-plane through-focus stack (defocus -310..310 nm), quasar(4,-pi/8) 0.4-0.8"),
}
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_TFLOPS = 157.3              # MI355X_MICROARCH.md: peak fp32 vector
# Algorithmic bytes per source-point*pixel: 40 for the fused pipeline =
# x-pass 24 (read mask window 8 + pupil window 8, write intermediate 8) + y-pass 16 (read
# intermediate 8, read-modify-write intensity 8).
ALGO_BYTES = {"xpass": 24.0, "ypass": 16.0}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3", choices=list(WORKLOADS))
    ap.add_argument("--shard", default=None, metavar="i/n",
                    help="single GPU only: process shard i of n of the source list (the per-rank work of the n-GPU run, "
                         "without the all-reduce)")
    ap.add_argument("--points", type=int, default=0,
                    help="profiling aid: only the first K consecutive source points (the JSON line is then marked partial)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def spawn_ranks(args):
    """Parent of a plain `python bench.py --gpus N`: start N fresh rank processes (this process has not touched the
    GPU and never will), relay rank 0's stdout, exit with the worst return code.  If one rank dies the others are
    ended (by PID) instead of waiting for a rendezvous that cannot complete."""
    import tempfile
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    worst = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0:
                worst = max(worst, abs(rc) or 1)
                for q in live:
                    q.terminate()
    out0.seek(0)
    sys.stdout.write(out0.read())
    sys.stdout.flush()
    sys.exit(worst)


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    # Test hooks (tests/test_gpu_bench_contract.py): on a box with ONE GPU the multi-rank code path of this file is
    # exercised with every rank on cuda:0 and a gloo group (RCCL refuses two ranks on one device).
    backend = os.environ.get("LITHO_BENCH_BACKEND", "nccl")
    share_gpu = os.environ.get("LITHO_BENCH_SHARE_GPU") == "1"
    dev = torch.device("cuda", 0 if share_gpu else local_rank)
    torch.cuda.set_device(dev)
    group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)      # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)
        group = dist.group.WORLD

    import lithographysimulator_amd as L
    from lithographysimulator_amd import _native as nat
    from lithographysimulator_amd.synthetic import bernoulli_mask

    import contextlib
    pn, skind, ab, planes, desc = WORKLOADS[args.workload]
    _notices = contextlib.redirect_stdout(sys.stderr)      # the object API prints reference-style notices
    _notices.__enter__()
    mask = L.Mask(bernoulli_mask(pn), PS, dev)
    maskFT = mask.fraunhofer(WL, True)
    epsilon, N = mask.calculateEpsilonN(mask.deltaK, PS, WL)
    if skind == "circ":
        bitmap = L.LightSource(0.0, 0.5, pn, NA, device=dev).generateAnnular()
    else:
        ls = L.LightSource(0.4, 0.8, pn, NA, device=dev)
        bitmap = ls.generateAnnular() if skind == "annular" else ls.generateQuasar(4, -math.pi / 8)
    if planes == 1:
        pupil = L.Pupil(pn, WL, NA, None if ab is None else torch.tensor(ab, dtype=torch.float16), dev).generatePupilFunction()
    else:
        pupil = L.throughFocusPupils(pn, WL, NA, torch.tensor(ab, dtype=torch.float16), [float(d) for d in DEFOCUS_NM[:planes]], dev)
    S_full = int(bitmap.sum())
    torch.cuda.synchronize()
    _notices.__exit__(None, None, None)

    # ---- what one step processes
    lo, hi, shard_note = 0, S_full, None
    if args.shard:
        if world > 1:
            sys.exit("--shard is a single-GPU option")
        i, n = (int(v) for v in args.shard.split("/"))
        from lithographysimulator_amd.distributed import shard_bounds
        lo, hi = shard_bounds(S_full, i, n)
        shard_note = f"shard {i}/{n} of the source list (per-rank work of the {n}-GPU run, no all-reduce)"
    if args.points > 0:
        hi = min(hi, lo + args.points)
        shard_note = (shard_note + "; " if shard_note else "") + f"PARTIAL: first {hi - lo} consecutive source points only"
    S = hi - lo
    partial = (lo, hi) != (0, S_full)

    def step():
        if not partial:
            return L.abbeImage(mask, maskFT, pupil, bitmap, PS, mask.deltaK, WL, True, dev, group=group)
        sh = L.sourceShifts(bitmap, pn)[lo:hi]              # the three lines abbeImage runs per rank
        return L.postProcess(L.abbeIntensity(maskFT, pupil, sh, N), epsilon)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        image = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    units = float(S) * pn * pn * planes                          # source-point*pixels per step, whole job
    value = units * args.steps / elapsed

    # ---- roofline leg: HIP-event time of each kernel class over one more (untimed) step of this rank's shard
    nat.set_profiling(True)
    step()
    torch.cuda.synchronize()
    prof = nat.last_profile()
    plan = nat.last_plan()
    nat.set_profiling(False)
    # measured device-copy and device-fill ceilings of THIS box (1 GiB, beyond the 256 MiB Infinity Cache): what the
    # memory system sustains for plain streams, to put next to the 8 TB/s spec the fractions are quoted against
    copy_gbs = fill_gbs = None
    try:
        a = torch.empty(1 << 28, dtype=torch.float32, device=dev); b = torch.empty_like(a)
        b.copy_(a); a.zero_(); torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        for _ in range(5): b.copy_(a)
        e1.record()
        for _ in range(5): a.zero_()
        e2.record(); torch.cuda.synchronize()
        copy_gbs = 5 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9      # read + write bytes
        fill_gbs = 5 * a.numel() * 4 / (e1.elapsed_time(e2) * 1e-3) / 1e9          # write bytes
        del a, b
    except Exception:
        pass
    lines = {"xpass": plan["box_rows"], "ypass": pn}            # lines transformed per T item
    n_exec = pn if plan.get("coarse_grid") else N                # coarse-grid path: pn-point transforms on the grid q = 2 v
    line_flops = 5.0 * n_exec * math.log2(n_exec)                # nominal FFT flops of one transformed line
    # the kernels' names as rocprofv3 prints them (profiles/*_kernel_stats.csv), from the plan the library reports
    l2, full = int(math.log2(n_exec)), ", true" if n_exec == pn else ""
    if prof["ypass_kernel"] == "k_ypass_wave":
        yname = ("k_ypass_pair<13, 8>" if l2 == 13 else f"k_ypass_wave<12, 8{full}>" if l2 == 12
                 else f"k_ypass_rect<{l2}, 8{full}>")
    else:
        yname = "k_ypass_acc"
    xname = {1: f"k_xpass_abbe<{l2}, {int(math.log2(n_exec // pn))}, true, 1>", 2: "k_xpass_split<13>",
             3: f"k_xpass_rect<{l2}{', true' if full else ''}>"}.get(plan.get("fused_xpass"), "k_xpass")
    kern = {}
    for k in ("xpass", "ypass"):
        launches = max(1, prof[f"{k}_launches"])
        items = prof[f"{k}_points"]                              # T items = source points x planes
        avg_ms = prof[f"{k}_ms"] / launches
        per_launch = items / launches
        flops = line_flops * lines[k] * per_launch
        algo_bytes = ALGO_BYTES[k] * pn * pn * per_launch
        kern[k] = {"avg_launch_ms": avg_ms, "launches": launches, "items_per_launch": per_launch,
                   "nominal_flops_per_launch": flops,
                   "achieved_TFLOPs": flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0,
                   "algorithmic_bytes_per_launch": algo_bytes,
                   "effective_GBs": algo_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0,
                   "total_ms": prof[f"{k}_ms"]}
    dom = max(kern, key=lambda k: kern[k]["total_ms"])
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")       # PMC-derived memory-side bytes (rocprofv3 --pmc)
    if os.path.exists(tpath):
        try:
            entry = json.load(open(tpath)).get(args.workload, {})
            per_item = entry.get(dom + "_bytes_per_item")
            if per_item is not None:
                traffic = per_item * kern[dom]["items_per_launch"]
                traffic_src = entry.get("source")
        except Exception:
            traffic = None
    both_ms = kern["xpass"]["total_ms"] + kern["ypass"]["total_ms"]
    both_flops = sum(kern[k]["nominal_flops_per_launch"] * kern[k]["launches"] for k in kern)
    eff40 = 40.0 * pn * pn * prof["ypass_points"] / (both_ms * 1e-3) / 1e9 if both_ms else 0.0
    dom_s = kern[dom]["avg_launch_ms"] * 1e-3
    roofline = {"bound": "valu", "kernel": yname if dom == "ypass" else xname,
                "achieved": kern[dom]["achieved_TFLOPs"], "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": kern[dom]["achieved_TFLOPs"] / VALU_PEAK_TFLOPS, "traffic": traffic,
                "avg_launch_ms": kern[dom]["avg_launch_ms"],
                "nominal_flops_per_launch": kern[dom]["nominal_flops_per_launch"],
                "note": "measured limiter is fp32 VALU issue (PMC: profiles/); achieved = nominal 5*N*log2(N) flops per "
                        "transformed line (pruned transforms execute fewer) / HIP-event launch time; traffic = PMC "
                        "memory-side bytes per launch of this workload",
                "pipeline": {"achieved": both_flops / (both_ms * 1e-3) / 1e12 if both_ms else 0.0,
                             "frac": both_flops / (both_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS if both_ms else 0.0,
                             "note": "x-pass + y-pass nominal flops over their summed kernel time"},
                "hbm": {"peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "measured_copy_ceiling": copy_gbs, "measured_fill_ceiling": fill_gbs,
                        "measured": traffic / dom_s / 1e9 if traffic and dom_s > 0 else None,
                        "measured_frac": traffic / dom_s / 1e9 / HBM_PEAK_GBS if traffic and dom_s > 0 else None,
                        "traffic_source": traffic_src,
                        "algorithmic_bytes_per_launch": kern[dom]["algorithmic_bytes_per_launch"],
                        "effective_kernel": kern[dom]["effective_GBs"],
                        "effective_40B": eff40, "effective_40B_over_peak": eff40 / HBM_PEAK_GBS,
                        "note": "effective_* divide the SURVEY 8d byte MODEL (16 B/unit y-pass, 24 B/unit x-pass, 40 B/unit "
                                "pipeline) by kernel time; they exceed what HBM could stream because most of those "
                                "bytes are never moved -- not a roofline fraction"},
                "kernels": kern}

    out = {"metric": "Abbe source-points x image-pixels per second", "value": value, "unit": "source-pt*px/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic",
           "config": {"workload": f"BASELINE {args.workload}: {desc}" + (f" [{shard_note}]" if shard_note else ""),
                      "pn": pn, "fft_n": N, "executed_fft_n": n_exec, "source_points": S, "source_points_full": S_full, "planes": planes,
                      "points_per_rank": math.ceil(S / world), "pixel_size": PS, "wavelength": WL, "NA": NA,
                      "parallelism": f"source-point shards x{world}, one all-reduce" if world > 1 else "single GPU",
                      "plan": plan, "image_shape": list(image.shape)},
           "roofline": roofline}

    # ---- CPU baseline leg: the oracle's op-chain port of the reference loop, rank 0, N = 1 only
    if world == 1 and not args.no_cpu_baseline:
        from oracle import abbe_oracle as O
        # 32 threads is the fastest setting for this op chain on the GPU box's 2 x EPYC 9575F (256 hw threads:
        # 8 -> 2.1e7, 32 -> 2.6e7, 256 -> 1.6e6 pt*px/s; scripts/cpu_threads_probe.py), so that is the baseline.
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        K = {256: 64, 1024: 16, 2048: 8, 4096: 4}.get(pn, 8)
        shifts = L.sourceShifts(bitmap, pn)
        sel = shifts[(torch.arange(K, device=dev) * S_full) // K].cpu()
        p_one = pupil if planes == 1 else pupil[planes // 2]
        m_cpu, p_cpu = maskFT.cpu(), p_one.cpu()
        O.abbe_raw(m_cpu, p_cpu, sel[:1], N)                        # warm-up
        times = []
        for _ in range(3):
            c0 = time.perf_counter()
            ref_raw = O.abbe_raw(m_cpu, p_cpu, sel, N)
            times.append(time.perf_counter() - c0)
        tmed = statistics.median(times)
        gpu_raw = L.abbeIntensity(maskFT, p_one, sel.to(dev), N).cpu()
        parity = float((gpu_raw - ref_raw).abs().max() / ref_raw.max())
        out["cpu_baseline"] = {"value": K * pn * pn / tmed, "unit": "source-pt*px/s", "cores": torch.get_num_threads(),
                               "kind": "port",
                               "sample": f"{K} source points strided through the {S_full}-point list, full {pn}x{pn} grid"
                                         + (f", plane {planes // 2} of {planes}" if planes > 1 else "") +
                                         f", 1 warm-up + 3 reps, median {tmed:.2f} s; oracle/abbe_oracle.py abbe_raw "
                                         "(torch-CPU roll/mul/pad/fftshift/ifft2/ifftshift/crop/abs2/add)",
                               "gpu_vs_cpu_rel_to_max": parity}

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
