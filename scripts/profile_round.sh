#!/bin/bash
# Round profile capture on the GPU box (two steps: a full run, copy traffic_*.json into profiles/traffic.json, then
# SKIP_PMC=1 so that the bench lines carry the traffic measured for THESE kernels): bench lines, rocprofv3 kernel-trace stats of the SAME default command, PMC
# passes over bench.py itself (separate counter-only runs).  Usage: scripts/profile_round.sh r02
R=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --no-cpu-baseline --no-extra > $O/bench_under_rocprof.json 2> $O/rocprof.err
# (configs 1, 2, 4-shard and 5 ride in bench.json's extra_workloads since round 3; their own full lines with roofline:)
python3 bench.py --workload cfg1 > $O/bench_cfg1.json 2>/dev/null
python3 bench.py --workload cfg2 > $O/bench_cfg2.json 2>/dev/null
python3 bench.py --workload cfg4 --shard 0/8 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_cfg4_shard0of8.json 2>/dev/null
# per-rank work of the other 8-GPU runs (the predicted 8-GPU step is built from these)
python3 bench.py --workload cfg3 --shard 0/8 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg3_shard0of8.json 2>/dev/null
python3 bench.py --workload cfg5 --shard 0/8 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_cfg5_shard0of8.json 2>/dev/null
# SKIP_PMC=1: bench lines and kernel stats only (e.g. after profiles/traffic.json has been refreshed from the PMC passes)
[ -n "$SKIP_PMC" ] || bash scripts/pmc_traffic.sh $R cfg3 504 cfg2 3312 cfg4 64 cfg5 120 cfg1 3233   # (cfg2: 69 x its 48-point batch; a list of at most 64 x the 51-item cap -- a shorter list is re-batched evenly, 51 x 3-point chunks at 2016 points: another geometry than the full run's, caught by bench.py's geometry key in round 5)
# keep the summaries small: the raw per-dispatch CSVs stay in gpurun_out
find $O -name "*_agent_info.csv" -delete
find $O/stats -name "*_kernel_trace.csv" -delete        # 80k rows per image: the stats csv is the summary that gets committed
ls -la $O
