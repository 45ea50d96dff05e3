#!/bin/bash
# PMC-measured memory-side traffic of bench.py's OWN kernels (consecutive source points, the bench's batching):
#   scripts/pmc_traffic.sh <round> <workload> <points> [more "workload points" pairs]
# Separate counter-only passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), then scripts/pmc_traffic.py folds
# them into gpurun_out/<round>/traffic_<workload>.json (copy into profiles/traffic.json after review).
R=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$R
while [ $# -ge 2 ]; do
  W=$1; K=$2; shift 2
  D=gpurun_out/$R/pmc_$W
  B="python3 bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --points $K"
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/fetch -o p -- $B > $D.fetch.json 2> $D.fetch.err
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/write -o p -- $B > $D.write.json 2> $D.write.err
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $D/sq1 -o p -- $B > /dev/null 2> $D.sq1.err
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --output-format csv -d $D/sq2 -o p -- $B > /dev/null 2> $D.sq2.err
  python3 scripts/pmc_traffic.py $D $D.fetch.json $W > gpurun_out/$R/traffic_$W.json 2> gpurun_out/$R/traffic_$W.err
  python3 scripts/pmc_summary.py $D > gpurun_out/$R/pmc_${W}_summary.txt 2>&1
  rm -rf $D/*/*/*_agent_info.csv
done
