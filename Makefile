# Builds the drop-in library without Python (the same commands criteria3d_amd/build.py runs).
#   make            -> criteria3d_amd/csrc/libsf3d_hip.so (C ABI, gfx950) + shim/libsoilFluxes3D_mi355x.so (soilFluxes3D::v2 symbols)
#   make v1         -> shim/libsoilFluxes3D_v1_mi355x.so (retired v1 names)
#   make oracle     -> test infrastructure (CPU restatement; + oracle/_ref when /root/reference is mounted)
#   make test       -> CPU test suite
HIPCC   ?= /opt/rocm/bin/hipcc
CXX     ?= g++
CSRC    := criteria3d_amd/csrc
HIPFLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-gpu-rdc -Wall -Wno-unused-function

.PHONY: all product product-fm shim v1 oracle test clean
all: product shim

product: $(CSRC)/libsf3d_hip.so
# every part of the translation unit (the .inc files hold nearly all kernel and host code) and every header is a prerequisite
$(CSRC)/libsf3d_hip.so: $(CSRC)/sf3d_solver.hip $(CSRC)/sf3d_api.cpp $(wildcard $(CSRC)/*.inc $(CSRC)/*.h) include/sf3d.h
	$(HIPCC) $(HIPFLAGS) -Iinclude -I$(CSRC) -x hip $(CSRC)/sf3d_solver.hip $(CSRC)/sf3d_api.cpp -o $@

# the same product with the 0.50-ulp elementary functions of rounds 1-4 instead of the C library's: the build the oracle's
# fast-math twin (make -C oracle oracle-fm) is a twin of; load it with SF3D_PRODUCT_LIB=build_variants/libsf3d_hip_fm.so SF3D_TEST_RTOL=1e-6
product-fm: build_variants/libsf3d_hip_fm.so
build_variants/libsf3d_hip_fm.so: $(CSRC)/sf3d_solver.hip $(CSRC)/sf3d_api.cpp $(wildcard $(CSRC)/*.inc $(CSRC)/*.h) include/sf3d.h
	mkdir -p build_variants
	$(HIPCC) $(HIPFLAGS) -DSF3D_LIBM_GLIBC=0 -Iinclude -I$(CSRC) -x hip $(CSRC)/sf3d_solver.hip $(CSRC)/sf3d_api.cpp -o $@

shim: shim/libsoilFluxes3D_mi355x.so
shim/libsoilFluxes3D_mi355x.so: shim/sf3d_cxx_shim.cpp shim/soilFluxes3D_api.h include/sf3d.h $(CSRC)/libsf3d_hip.so
	$(CXX) -std=c++17 -O2 -fPIC -shared -Iinclude -Ishim $< -o $@ -L$(CSRC) -lsf3d_hip -Wl,-rpath,$(abspath $(CSRC))

v1: shim/libsoilFluxes3D_v1_mi355x.so
shim/libsoilFluxes3D_v1_mi355x.so: shim/sf3d_v1_alias.cpp shim/soilFluxes3D_v1_api.h include/sf3d.h $(CSRC)/libsf3d_hip.so
	$(CXX) -std=c++17 -O2 -fPIC -shared -Iinclude -Ishim $< -o $@ -L$(CSRC) -lsf3d_hip -Wl,-rpath,$(abspath $(CSRC))

oracle:
	$(MAKE) -C oracle oracle ref

test: all oracle
	python -m pytest tests -x -q -m "not gpu"

clean:
	rm -f $(CSRC)/libsf3d_hip.so shim/*.so shim/v1_alias_demo shim/v2_caller_demo
	$(MAKE) -C oracle clean
