#!/bin/bash
# Round profile capture on the GPU box: bench line, rocprofv3 kernel-trace stats of the SAME command,
# PMC passes (separate runs, counters only).  Usage: scripts/profile_round.sh r01
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$R
python3 bench.py > gpurun_out/$R/bench.json 2> gpurun_out/$R/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/stats -o bench -- python3 bench.py --no-cpu-baseline > gpurun_out/$R/bench_under_rocprof.json 2> gpurun_out/$R/rocprof.err
for w in cfg2; do python3 bench.py --workload $w --no-cpu-baseline > gpurun_out/$R/bench_$w.json 2>/dev/null; done
python3 scripts/quick_time.py 4096 256 > gpurun_out/$R/quick_4096.txt 2>/dev/null
P="python3 scripts/quick_time.py 2048 510"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/$R/pmc1 -o p -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --output-format csv -d gpurun_out/$R/pmc2 -o p -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/$R/pmc3 -o p -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$R/pmc4 -o p -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/$R/pmc5 -o p -- $P > /dev/null 2>&1
python3 scripts/pmc_summary.py gpurun_out/$R > gpurun_out/$R/pmc_summary.txt 2>&1
find gpurun_out/$R -name "*.csv" -size +2M -delete
ls -R gpurun_out/$R | head -40
