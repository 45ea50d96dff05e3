#!/bin/bash
# A/B counter capture for one workload under two settings of an environment knob:
#   scripts/pmc_ab.sh <out> <workload> <points> <KNOB> <value A> <value B> [counter groups...]
# each counter group is one --pmc pass (separate runs, counters only).
O=gpurun_out/$1; W=$2; P=$3; K=$4; A=$5; B=$6; shift 6
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $O
i=0
for grp in "$@"; do
  for v in $A $B; do
    export $K=$v
    rocprofv3 --pmc $grp --output-format csv -d $O/${K}_${v}_g$i -o pmc -- python3 bench.py --workload $W --points $P --steps 1 --warmup 0 --no-cpu-baseline --no-extra > /dev/null 2> $O/err_${v}_$i.txt
  done
  i=$((i+1))
done
python3 scripts/pmc_summary.py $O > $O/summary.txt
cat $O/summary.txt
