#!/bin/bash
# Memory-path counters (TA / TCP / TCC / TD) of bench.py's own kernels, one counter-only pass per group:
# (every pass under `timeout`: a counter set the hardware cannot schedule makes rocprofv3 abort and then hang)
#   scripts/pmc_mempath.sh <round> <workload> <points> [more pairs]
R=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$R
while [ $# -ge 2 ]; do
  W=$1; K=$2; shift 2
  D=gpurun_out/$R/mem_$W
  B="python3 bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --points $K"
  timeout 180 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $D/ta -o p -- $B > /dev/null 2> $D.ta.err
  timeout 180 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $D/tcp -o p -- $B > /dev/null 2> $D.tcp.err
  timeout 180 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $D/tcc -o p -- $B > /dev/null 2> $D.tcc.err
  timeout 180 rocprofv3 --kernel-trace --pmc TD_TD_BUSY_sum TD_TC_STALL_sum --output-format csv -d $D/td -o p -- $B > /dev/null 2> $D.td.err
  python3 scripts/pmc_summary.py $D > gpurun_out/$R/mem_${W}_summary.txt 2>&1
  rm -rf $D/*/*/*_agent_info.csv
done
