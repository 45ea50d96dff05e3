#!/bin/bash
# A/B builds of one FFT size: each variant relinks the library with inst_$L2 rebuilt under other flags.
#   scripts/build_variants.sh 12 name1 "flags1" name2 "flags2" ...
set -e
cd "$(dirname "$0")/.."
L2=$1; shift
mkdir -p build/variants
OBJS="build/abbe_engine.o build/optics.o build/common.o $(ls build/inst_*.o | grep -v inst_$L2)"
build() {
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-signed-zeros -fno-slp-vectorize $2 -c lithographysimulator_amd/csrc/inst_$L2.hip -o build/variants/inst_${L2}_$1.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib_$1.so $OBJS build/variants/inst_${L2}_$1.o
}
while [ $# -gt 0 ]; do build "$1" "$2" & shift 2; done
wait
ls build/variants/*.so
