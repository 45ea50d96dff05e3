#!/bin/bash
# A/B builds of one FFT size: each variant relinks the library with inst_$L2 rebuilt under other flags.
#   scripts/build_variants.sh 12 name1 "flags1" name2 "flags2" ...      (w11: the wave-level unit instw_11 instead)
# Output: build/variants/lib_<name>.so (never the product path).  Flags containing LITHO_DIAG_ give a
# diagnostic build (wrong results, timing only): LITHO_DIAG_BUILD is added and the library reports
# litho_target_arch() == "gfx950-diag" so that the binding refuses it unless LITHO_ALLOW_DIAG=1.
set -e
cd "$(dirname "$0")/.."
L2=$1; shift
UNIT=inst_$L2
case "$L2" in w*) UNIT=instw_${L2#w};; esac
mkdir -p build/variants
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
OBJS="build/abbe_engine.o build/optics.o build/layout.o build/plan_dry_run.o $(ls build/inst_*.o build/instw_*.o | grep -v "/$UNIT.o")"
hipcc $BASE -DLITHO_DIAG_BUILD -c lithographysimulator_amd/csrc/common.hip -o build/variants/common_diag.o
build() {
  local extra="" common="build/common.o"
  case "$2" in *LITHO_DIAG_*) extra="-DLITHO_DIAG_BUILD"; common="build/variants/common_diag.o";; esac
  hipcc $BASE -fno-signed-zeros -fno-slp-vectorize $extra $2 -c lithographysimulator_amd/csrc/$UNIT.hip -o build/variants/${UNIT}_$1.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib_$1.so $OBJS $common build/variants/${UNIT}_$1.o
}
while [ $# -gt 0 ]; do build "$1" "$2" & shift 2; done
wait
ls build/variants/*.so
