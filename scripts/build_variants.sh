#!/bin/bash
# A/B builds of the N=4096 kernels: each variant relinks the library with inst_12 rebuilt under other flags.
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
OBJS="build/abbe_engine.o build/optics.o build/common.o $(ls build/inst_*.o | grep -v inst_12)"
build() { # name, flags
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $2 -c lithographysimulator_amd/csrc/inst_12.hip -o build/variants/inst_12_$1.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib_$1.so $OBJS build/variants/inst_12_$1.o
}
build base "-fno-signed-zeros" &
build fullbar "-fno-signed-zeros -DLITHO_FULL_BARRIER" &
build nopref "-fno-signed-zeros -DLITHO_NO_PREFETCH" &
build nonsz "" &
wait
build wg1 "-fno-signed-zeros -DLITHO_WG_PER_CU=1" &
build nopref_fullbar "-fno-signed-zeros -DLITHO_NO_PREFETCH -DLITHO_FULL_BARRIER" &
wait
ls -la build/variants/*.so
