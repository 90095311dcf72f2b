# Builds the product library (HIP kernels + C ABI + host mirror) for gfx950, the oracle, and
# the headless driver.  `python -c "import __graft_entry__ as g; g.build()"` drives this.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := path_tracer_amd/csrc
HOST    := path_tracer_amd/host
LIBDIR  := path_tracer_amd/lib
LIB     := $(LIBDIR)/libpt_amd.so
# -fno-slp-vectorize: hipcc otherwise packs adjacent scalar f32 ops into v_pk_* plus v_mov shuffles;
# v_pk_fma_f32 issues at half the rate of v_fma_f32 on gfx950, so packing only adds instructions
# (measured +15..20 % Msamples/s without it, profiles/r1_variants.json)
KFLAGS  ?= -fno-slp-vectorize
HIPFLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -Iinclude
OBJS    := build/pt_api.o build/pt_path_kernel.o build/pt_f64_kernels.o build/pt_builtin_scenes.o build/pt_fly_camera.o

all: $(LIB) oracle tools/pt_headless

$(LIB): $(OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

build/pt_api.o: $(CSRC)/pt_api.cpp $(CSRC)/pt_device.h $(CSRC)/pt_error.h include/pt.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

build/pt_path_kernel.o: $(CSRC)/pt_path_kernel.hip $(CSRC)/pt_device.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) $(KFLAGS) -c $< -o $@

# FP64 kernels: no FMA contraction, so every operation is one IEEE double op (bit-exact ids)
build/pt_f64_kernels.o: $(CSRC)/pt_f64_kernels.hip $(CSRC)/pt_device.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -ffp-contract=off -c $< -o $@

build/pt_builtin_scenes.o: $(HOST)/pt_builtin_scenes.cpp $(CSRC)/pt_error.h include/pt_host.hpp include/pt.h
	@mkdir -p build
	g++ -O2 -fPIC -std=c++17 -Wall -ffp-contract=off -Iinclude -c $< -o $@

build/pt_fly_camera.o: $(HOST)/pt_fly_camera.cpp $(CSRC)/pt_error.h include/pt.h
	@mkdir -p build
	g++ -O2 -fPIC -std=c++17 -Wall -ffp-contract=off -Iinclude -c $< -o $@

oracle:
	$(MAKE) -C oracle

tools/pt_headless: tools/pt_headless.cpp include/pt_host.hpp include/pt.h $(LIB)
	g++ -O2 -std=c++17 -Wall -Iinclude -o $@ tools/pt_headless.cpp -L$(LIBDIR) -lpt_amd -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)'

isa: $(CSRC)/pt_path_kernel.hip $(CSRC)/pt_device.h
	@mkdir -p build/isa
	$(HIPCC) $(HIPFLAGS) $(KFLAGS) -save-temps=obj -Rpass-analysis=kernel-resource-usage -c $(CSRC)/pt_path_kernel.hip -o build/isa/pt_path_kernel.o 2> build/isa/resource_usage.txt || true

clean:
	rm -rf build $(LIB) tools/pt_headless
	$(MAKE) -C oracle clean

.PHONY: all oracle clean isa

# kernel-variant build for A/B runs:  make variant NAME=v1 VFLAGS="-DPT_BLOCK=128"
# (VFLAGS are added to KFLAGS; overriding KFLAGS itself drops -fno-slp-vectorize and costs ~13 %)
variant: build/pt_api.o build/pt_f64_kernels.o build/pt_builtin_scenes.o build/pt_fly_camera.o
	@mkdir -p build/var_$(NAME)
	$(HIPCC) $(HIPFLAGS) $(KFLAGS) $(VFLAGS) -c $(CSRC)/pt_path_kernel.hip -o build/var_$(NAME)/pt_path_kernel.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $(LIBDIR)/libpt_amd_$(NAME).so build/pt_api.o build/var_$(NAME)/pt_path_kernel.o build/pt_f64_kernels.o build/pt_builtin_scenes.o build/pt_fly_camera.o
