#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_wave; rm -rf $OUT; mkdir -p $OUT
export LITHO_ABBE_W64_8192=1
for cfg in "2048 68" "4096 32"; do set -- $cfg
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/a$1 -o p -- python3 scripts/quick_time.py $1 $2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR --output-format csv -d $OUT/b$1 -o p -- python3 scripts/quick_time.py $1 $2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/c$1 -o p -- python3 scripts/quick_time.py $1 $2 > /dev/null 2>&1
done
python3 scripts/pmc_summary.py $OUT | grep -v "field\|RealImage\|xpass"
