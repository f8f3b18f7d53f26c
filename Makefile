# Builds libhb.so (C-ABI + HIP kernels for gfx950), the host tools and the CPU oracle.
HIPCC ?= hipcc
ARCH ?= gfx950
CSRC := humanoid_mujoco_amd/csrc
HOST_SRCS := $(CSRC)/hb_api.cpp $(CSRC)/mjcf.cpp $(CSRC)/setconst.cpp $(CSRC)/model_io.cpp $(CSRC)/mesh.cpp
HIP_SRCS := $(CSRC)/hb_step.hip $(CSRC)/hb_step_duo.hip $(CSRC)/hb_narrow.hip $(CSRC)/hb_env.hip
HDRS := $(wildcard $(CSRC)/*.hpp) include/hb.h
LIB := humanoid_mujoco_amd/libhb.so
FLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-result -ffp-contract=on -fno-slp-vectorize -fno-vectorize $(EXTRA)

all: $(LIB) build/hb_compile build/hb_testspeed oracle

# host sources are plain C++ (HIP runtime API only); only the kernels are compiled for the device
HOSTFLAGS := -O2 -std=c++17 -fPIC -Wall -Wno-unused-result -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include
HOST_OBJS := $(patsubst $(CSRC)/%.cpp,build/obj/%.o,$(HOST_SRCS))

build/obj/%.o: $(CSRC)/%.cpp $(HDRS)
	@mkdir -p build/obj
	g++ $(HOSTFLAGS) -c $< -o $@

# one object per kernel translation unit (make -j builds them side by side; an edit rebuilds its own unit only)
HIP_OBJS := $(patsubst $(CSRC)/%.hip,build/obj/%.o,$(HIP_SRCS))
build/obj/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build/obj
	$(HIPCC) $(FLAGS) -c -x hip $< -o $@

$(LIB): $(HOST_OBJS) $(HIP_OBJS)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $^

# diagnostic build with per-phase cycle stamps (never used for timing or by the product)
build/libhb_stamps.so: $(HOST_SRCS) $(HIP_SRCS) $(HDRS)
	@mkdir -p build/obj_stamps
	for f in $(HOST_SRCS); do g++ $(HOSTFLAGS) -DHB_STAMPS -c $$f -o build/obj_stamps/$$(basename $$f .cpp).o || exit 1; done
	for f in $(HIP_SRCS); do $(HIPCC) $(FLAGS) -DHB_STAMPS -c -x hip $$f -o build/obj_stamps/$$(basename $$f .hip).o || exit 1; done
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ build/obj_stamps/*.o

# diagnostic build: section cycles inside the Newton iterations (tools/gpu_newton_bench.py --probe)
build/libhb_probe.so: $(HOST_SRCS) $(HIP_SRCS) $(HDRS)
	@mkdir -p build/obj_probe
	for f in $(HOST_SRCS); do g++ $(HOSTFLAGS) -DHB_STAMPS -c $$f -o build/obj_probe/$$(basename $$f .cpp).o || exit 1; done
	for f in $(HIP_SRCS); do $(HIPCC) $(FLAGS) -DHB_STAMPS -DHB_PROBE_NEWTON -c -x hip $$f -o build/obj_probe/$$(basename $$f .hip).o || exit 1; done
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ build/obj_probe/*.o

build/hb_compile: tools/hb_compile.cpp $(CSRC)/mjcf.cpp $(CSRC)/setconst.cpp $(CSRC)/model_io.cpp $(CSRC)/mesh.cpp $(HDRS)
	@mkdir -p build
	g++ -O2 -std=c++17 -Wall -o $@ tools/hb_compile.cpp $(CSRC)/mjcf.cpp $(CSRC)/setconst.cpp $(CSRC)/model_io.cpp $(CSRC)/mesh.cpp

build/hb_testspeed: tools/hb_testspeed.cpp $(LIB) include/hb.h
	@mkdir -p build
	g++ -O2 -std=c++17 -Wall -Iinclude -o $@ tools/hb_testspeed.cpp -Lhumanoid_mujoco_amd -lhb -Wl,-rpath,'$$ORIGIN/../humanoid_mujoco_amd'

# the same host program against the diagnostic library: prints the per-stage table (testspeed.cc:235-288)
build/hb_testspeed_stamps: tools/hb_testspeed.cpp build/libhb_stamps.so include/hb.h
	g++ -O2 -std=c++17 -Wall -Iinclude -o $@ tools/hb_testspeed.cpp -Lbuild -l:libhb_stamps.so -Wl,-rpath,'$$ORIGIN'

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(LIB) build/hb_compile build/hb_testspeed
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
