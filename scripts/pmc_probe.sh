#!/bin/bash
# PMC counters for the two pass kernels on a short run (separate passes; no trace domains).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_$1
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/p1 -o p -- python3 scripts/quick_time.py ${2:-2048} ${3:-252} > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --output-format csv -d $OUT/p2 -o p -- python3 scripts/quick_time.py ${2:-2048} ${3:-252} > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p3 -o p -- python3 scripts/quick_time.py ${2:-2048} ${3:-252} > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/p4 -o p -- python3 scripts/quick_time.py ${2:-2048} ${3:-252} > $OUT/p4.log 2>&1
ls $OUT/*
