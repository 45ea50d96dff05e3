# Builds liblitho_abbe.so (hand-written HIP for gfx950 behind a C ABI) and the C oracle.
HIPCC ?= hipcc
ARCH ?= gfx950
CSRC := lithographysimulator_amd/csrc
OUT := lithographysimulator_amd/lib/liblitho_abbe.so
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function
# FFT kernels: no signed zeros, so that literal-zero input slots fold through the butterflies.
# No SLP packing: v_pk_*_f32 are not faster than two scalar ops on gfx950 and cost ~100 register moves
# per transform (and 40-80 VGPRs).
FFTFLAGS := -fno-signed-zeros -fno-slp-vectorize
# Wave-level kernels live in their own translation units (instw_*.hip) so that they can take their own scheduling
# strategy.  Same-box A/B (us per source point, default vs max-ilp): k_ypass_wave<12> 9.21 -> 9.03, but its full-output
# variant (the coarse-grid path at 4096^2) spills 73 registers under max-ilp against 3 -- so: default everywhere.
WAVEFLAGS_12 ?=
# instw_11 (k_ypass_rect<11, ...>: configs 3 and 5): the transposes' 16 LDS read bases re-derived per line instead of living in
# registers across the loop (LITHO_TRANSPOSE_OPAQUE, wave_fft.hpp) -- bit-identical, y-pass 4.60 / 4.64 -> 4.54 / 4.55 us per
# source point in two alternating A/B pairs on one box (round 5); 1024^2 is indifferent (1.00 / 1.01 vs 1.01 / 0.99), and at
# instw_12 the same switch makes k_ypass_wave<12, 4, true> spill 54 registers: 11 only.
WAVEFLAGS_11 ?= -DLITHO_TRANSPOSE_OPAQUE
# and the N = 8192 split x-pass: 28.5 -> 27.0 us per source point at 4096^2 (the N = 4096 x-pass prefers the default)
INSTFLAGS_13 ?= -mllvm -amdgpu-sched-strategy=max-ilp
INST := $(patsubst $(CSRC)/%.hip,build/%.o,$(wildcard $(CSRC)/inst_*.hip))
INSTW := $(patsubst $(CSRC)/%.hip,build/%.o,$(wildcard $(CSRC)/instw_*.hip))
HDRS := $(CSRC)/abbe_plan.hpp $(CSRC)/fft_core.hpp $(CSRC)/wave_fft.hpp $(CSRC)/engine_kernels.hpp $(CSRC)/wave_kernels.hpp $(CSRC)/engine_common.hpp include/litho_abbe.h

all: $(OUT) oracle

build/inst_%.o: $(CSRC)/inst_%.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) $(FFTFLAGS) $(INSTFLAGS_$*) -c $< -o $@
build/instw_%.o: $(CSRC)/instw_%.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) $(FFTFLAGS) $(WAVEFLAGS_$*) -c $< -o $@
build/abbe_engine.o: $(CSRC)/abbe_engine.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
build/optics.o: $(CSRC)/optics.hip $(CSRC)/engine_common.hpp include/litho_abbe.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -ffp-contract=off -c $< -o $@
build/layout.o: $(CSRC)/layout.hip $(CSRC)/engine_common.hpp include/litho_abbe.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -ffp-contract=off -c $< -o $@
build/common.o: $(CSRC)/common.hip $(CSRC)/engine_common.hpp include/litho_abbe.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
# the host-only planner's dry-run entry point: plain C++ (g++, no HIP header) -- tests/test_planner_cpu.py builds it alone too
build/plan_dry_run.o: $(CSRC)/plan_dry_run.cpp $(CSRC)/abbe_plan.hpp include/litho_abbe.h
	@mkdir -p build
	g++ -O2 -std=c++17 -fPIC -Wall -Wextra -c $< -o $@
$(OUT): build/abbe_engine.o build/optics.o build/layout.o build/common.o build/plan_dry_run.o $(INST) $(INSTW)
	@mkdir -p lithographysimulator_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^

oracle:
	$(MAKE) -C oracle

# the C ABI driven from plain C (no Python, no torch): the reference's demo configuration; build/c_abi_demo [pn]
examples: build/c_abi_demo
build/c_abi_demo: examples/c_abi_demo.c include/litho_abbe.h $(OUT)
	@mkdir -p build
	gcc -std=c99 -Wall -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude $< -Llithographysimulator_amd/lib -llitho_abbe -L/opt/rocm/lib -lamdhip64 -lm -o $@

clean:
	rm -rf build $(OUT)
	$(MAKE) -C oracle clean
.PHONY: all oracle clean examples
