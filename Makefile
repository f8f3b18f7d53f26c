# Builds libhb.so (C-ABI + HIP kernels for gfx950), the host tools and the CPU oracle.
HIPCC ?= hipcc
ARCH ?= gfx950
CSRC := humanoid_mujoco_amd/csrc
HOST_SRCS := $(CSRC)/hb_api.cpp $(CSRC)/mjcf.cpp $(CSRC)/setconst.cpp $(CSRC)/model_io.cpp
HIP_SRCS := $(CSRC)/hb_kernels.hip
HDRS := $(wildcard $(CSRC)/*.hpp) include/hb.h
LIB := humanoid_mujoco_amd/libhb.so
FLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-result

all: $(LIB) build/hb_compile build/hb_testspeed oracle

$(LIB): $(HOST_SRCS) $(HIP_SRCS) $(HDRS)
	$(HIPCC) $(FLAGS) -shared -o $@ -x hip $(HOST_SRCS) $(HIP_SRCS)

build/hb_compile: tools/hb_compile.cpp $(CSRC)/mjcf.cpp $(CSRC)/setconst.cpp $(CSRC)/model_io.cpp $(HDRS)
	@mkdir -p build
	g++ -O2 -std=c++17 -Wall -o $@ tools/hb_compile.cpp $(CSRC)/mjcf.cpp $(CSRC)/setconst.cpp $(CSRC)/model_io.cpp

build/hb_testspeed: tools/hb_testspeed.cpp $(LIB) include/hb.h
	@mkdir -p build
	g++ -O2 -std=c++17 -Wall -Iinclude -o $@ tools/hb_testspeed.cpp -Lhumanoid_mujoco_amd -lhb -Wl,-rpath,'$$ORIGIN/../humanoid_mujoco_amd'

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(LIB) build/hb_compile build/hb_testspeed
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
