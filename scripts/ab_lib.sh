#!/bin/bash
# A/B of two builds of the library on ONE box, alternating: scripts/ab_lib.sh <other.so> <workload> <points> [reps]
# (LITHO_ABBE_LIB selects the library; the product build is the default)
OTHER=$1; W=$2; P=$3; R=${4:-2}
for r in $(seq $R); do
  for lib in product other; do
    if [ $lib = product ]; then unset LITHO_ABBE_LIB; else export LITHO_ABBE_LIB=$PWD/$OTHER; fi
    python3 bench.py --workload $W --points $P --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /tmp/ab.json 2>/dev/null
    echo "$W $lib: $(python3 scripts/show_bench.py /tmp/ab.json | sed -n '1p' | cut -c1-60) | $(python3 scripts/show_bench.py /tmp/ab.json | sed -n '3p' | awk '{print $3, $4}') | $(python3 scripts/show_bench.py /tmp/ab.json | sed -n '4p' | awk '{print $3, $4}')"
  done
done
