cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
export LITHO_ABBE_COARSE=2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/cfg1c -o c -- python3 bench.py --workload cfg1 --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/r04/cfg1c.json 2> gpurun_out/r04/cfg1c.err
find gpurun_out/r04/cfg1c -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-200 | head -40
unset LITHO_ABBE_COARSE
python3 bench.py --workload cfg1 --steps 20 --warmup 2 --no-cpu-baseline | python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('direct', r['ms_per_step'], r['value'])"
LITHO_ABBE_COARSE=2 python3 bench.py --workload cfg1 --steps 20 --warmup 2 --no-cpu-baseline | python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('coarse', r['ms_per_step'], r['value'])"
find gpurun_out/r04/cfg1c -name "*_kernel_trace.csv" -delete; find gpurun_out/r04/cfg1c -name "*agent_info.csv" -delete
