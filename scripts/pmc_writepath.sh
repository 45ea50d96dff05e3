#!/bin/bash
# L2 -> fabric write-path counters of bench.py's own kernels (two counters per pass, every pass under `timeout`):
#   scripts/pmc_writepath.sh <round> <workload> <points>
R=$1; W=$2; K=$3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=gpurun_out/$R/wr_$W
mkdir -p $D
B="python3 bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --points $K"
i=0
for C in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
         "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_LEVEL_sum" "TCC_BUSY_sum TCC_TAG_STALL_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_WRREQ_DRAM_sum TCC_NORMAL_WRITEBACK_sum" "GRBM_GUI_ACTIVE TCC_IB_STALL_sum" \
         "TA_TA_BUSY_sum" "TA_BUFFER_WRITE_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum" "TA_BUFFER_TOTAL_CYCLES_sum" \
         "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TD_TD_BUSY_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  timeout 180 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D/p$i -o p -- $B > /dev/null 2> $D.p$i.err || echo "pass $i failed: $C"
done
python3 scripts/pmc_summary.py $D > gpurun_out/$R/wr_${W}_summary.txt 2>&1
rm -rf $D/*/*/*_agent_info.csv
