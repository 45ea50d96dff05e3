#!/bin/bash
# Alternating A/B timing on ONE box: scripts/ab_compare.sh <pn> <planes> <K> <reps> "ENV_A" "ENV_B" ...
pn=$1; planes=$2; K=$3; reps=$4; shift 4
hostname
for r in $(seq 1 $reps); do
  for e in "$@"; do
    printf "%-44s " "$e"
    env $e python scripts/stack_time.py $pn $planes $K 2>&1 | tail -1 | cut -c30-100
  done
done
