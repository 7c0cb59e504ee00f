# Build libpmx_hip.so (HIP kernels + C ABI) for gfx950, and the CPU oracle (test infrastructure).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := pharmsol_amd/csrc
LIB := pharmsol_amd/lib/libpmx_hip.so
# -ffp-contract=off on the HOST side: the population compiler must evaluate covariate
# lines (slope*t + intercept) exactly like the reference; device code keeps FMA contraction.
HOSTFLAGS := -O2 -std=c++17 -fPIC -Wall -Wextra -ffp-contract=off -Iinclude
DEVFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-parameter -Iinclude
OBJ := $(CSRC)/build/pmx_compile.o $(CSRC)/build/pmx_api.o $(CSRC)/build/pmx_kernels.o $(CSRC)/build/pmx_jit.o $(CSRC)/build/pmx_alloc.o $(CSRC)/build/pmx_shard.o
DEVHDR := $(CSRC)/pmx_devtypes.hpp $(CSRC)/pmx_device.hpp $(CSRC)/pmx_ode.hpp $(CSRC)/pmx_structures.hpp $(CSRC)/pmx_userlag.hpp $(CSRC)/pmx_analytical.hpp $(CSRC)/pmx_ode_user.hpp include/pmx.h

all: $(LIB) oracle

$(CSRC)/build/pmx_compile.o: $(CSRC)/pmx_compile.cpp $(CSRC)/pmx_compile.hpp $(CSRC)/pmx_devtypes.hpp include/pmx.h
	@mkdir -p $(CSRC)/build
	g++ $(HOSTFLAGS) -c $< -o $@

$(CSRC)/build/pmx_api.o: $(CSRC)/pmx_api.cpp $(CSRC)/pmx_compile.hpp $(CSRC)/pmx_kernels.hpp $(CSRC)/pmx_structures.hpp $(CSRC)/pmx_jit.hpp $(DEVHDR)
	@mkdir -p $(CSRC)/build
	$(HIPCC) $(DEVFLAGS) -ffp-contract=off -x hip -c $< -o $@

$(CSRC)/build/pmx_kernels.o: $(CSRC)/pmx_kernels.hip $(CSRC)/pmx_kernels.hpp $(CSRC)/pmx_structures.hpp $(CSRC)/pmx_compile.hpp $(DEVHDR)
	@mkdir -p $(CSRC)/build
	$(HIPCC) $(DEVFLAGS) -c $< -o $@

# the device headers a run-time (hiprtc) compile of a user model includes, embedded as string literals
$(CSRC)/build/pmx_jit_headers.inc: $(DEVHDR) tools/embed_headers.py
	@mkdir -p $(CSRC)/build
	python3 tools/embed_headers.py $@ include/pmx.h $(CSRC)/pmx_devtypes.hpp $(CSRC)/pmx_device.hpp $(CSRC)/pmx_ode.hpp $(CSRC)/pmx_structures.hpp $(CSRC)/pmx_userlag.hpp $(CSRC)/pmx_analytical.hpp $(CSRC)/pmx_ode_user.hpp

$(CSRC)/build/pmx_alloc.o: $(CSRC)/pmx_alloc.cpp include/pmx.h
	@mkdir -p $(CSRC)/build
	$(HIPCC) $(DEVFLAGS) -x hip -c $< -o $@

$(CSRC)/build/pmx_shard.o: $(CSRC)/pmx_shard.cpp include/pmx.h
	@mkdir -p $(CSRC)/build
	$(HIPCC) $(DEVFLAGS) -x hip -c $< -o $@

$(CSRC)/build/pmx_jit.o: $(CSRC)/pmx_jit.cpp $(CSRC)/pmx_jit.hpp $(CSRC)/build/pmx_jit_headers.inc
	$(HIPCC) $(DEVFLAGS) -I$(CSRC)/build -x hip -c $< -o $@

$(LIB): $(OBJ)
	@mkdir -p pharmsol_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJ) -L/opt/rocm/lib -lhiprtc -ldl

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(CSRC)/build pharmsol_amd/lib oracle/_build

.PHONY: all oracle clean

# C++ host-facade test binary (links the product library and, as the checker, the CPU oracle)
tests/cpp/facade_test: tests/cpp/facade_test.cpp include/pharmsol_hip.hpp include/pmx.h $(LIB) oracle
	g++ -O1 -std=c++17 -Wall -Wextra -o $@ tests/cpp/facade_test.cpp -Lpharmsol_amd/lib -lpmx_hip -Loracle/_build -lpmx_oracle \
	    -Wl,-rpath,'$$ORIGIN/../../pharmsol_amd/lib' -Wl,-rpath,'$$ORIGIN/../../oracle/_build' -Wl,-rpath,/opt/rocm/lib
