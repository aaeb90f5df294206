# Builds, in-tree:
#   raytracer_amd/lib/libmirt_host.so  host scene model (g++, no GPU code)
#   raytracer_amd/lib/libmirt.so       HIP kernels + C ABI for gfx950 (hipcc)
#   raytracer_amd/lib/libmirt.a        the same objects as a static archive
#   oracle/liboracle.so                CPU oracle (gcc) — test infrastructure only
#   raytracer_amd/lib/render_c         examples/render_c.c, a plain-C user of the two ABIs
#
# Floating-point contract: no implicit contraction anywhere (-ffp-contract=off);
# fused multiply-adds appear only where the source says fmaf(), correctly
# rounded division and square root on the device.

HIPCC ?= /opt/rocm/bin/hipcc
CXX   ?= g++
CC    ?= gcc
ARCH  ?= gfx950

LIBDIR := raytracer_amd/lib
HOST_SRC := $(wildcard raytracer_amd/csrc/host/*.cpp)
HOST_HDR := $(wildcard raytracer_amd/csrc/host/*.hpp) include/mirt.h include/mirt_host.h
DEV_SRC  := $(wildcard raytracer_amd/csrc/device/*.hip)
DEV_HDR  := $(wildcard raytracer_amd/csrc/device/*.h) include/mirt.h
DEV_OBJ  := $(patsubst raytracer_amd/csrc/device/%.hip,build/device/%.o,$(DEV_SRC))

HOST_FLAGS := -std=c++17 -O2 -fPIC -ffp-contract=off -Wall -Wextra -Iinclude -pthread
DEV_FLAGS  := --offload-arch=$(ARCH) -std=c++17 -O3 -fPIC -ffp-contract=off \
              -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc -Wall -Iinclude
ORACLE_FLAGS := -std=c11 -O2 -fPIC -ffp-contract=off -mfma -fopenmp -Wall -Wextra -Iinclude

all: host device oracle examples

host: $(LIBDIR)/libmirt_host.so
device: $(LIBDIR)/libmirt.so $(LIBDIR)/libmirt.a
oracle: oracle/liboracle.so
examples: $(LIBDIR)/render_c

$(LIBDIR)/libmirt_host.so: $(HOST_SRC) $(HOST_HDR)
	@mkdir -p $(LIBDIR)
	$(CXX) $(HOST_FLAGS) -shared -o $@ $(HOST_SRC)

build/device/%.o: raytracer_amd/csrc/device/%.hip $(DEV_HDR)
	@mkdir -p build/device
	$(HIPCC) $(DEV_FLAGS) -c $< -o $@

$(LIBDIR)/libmirt.so: $(DEV_OBJ)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(DEV_OBJ)

$(LIBDIR)/libmirt.a: $(DEV_OBJ)
	@mkdir -p $(LIBDIR)
	rm -f $@ && ar rcs $@ $(DEV_OBJ)

oracle/liboracle.so: oracle/oracle.c oracle/oracle_classic.inc oracle/oracle.h include/mirt.h
	$(CC) $(ORACLE_FLAGS) -shared -o $@ oracle/oracle.c -lm

# plain-C user of both ABIs (finds the libraries next to itself)
$(LIBDIR)/render_c: examples/render_c.c include/mirt.h include/mirt_host.h $(LIBDIR)/libmirt.so $(LIBDIR)/libmirt_host.so
	$(CC) -std=c11 -O2 -Wall -Wextra -Iinclude $< -L$(LIBDIR) -lmirt -lmirt_host -Wl,-rpath,'$$ORIGIN' -o $@

clean:
	rm -rf build $(LIBDIR)/*.so $(LIBDIR)/*.a $(LIBDIR)/render_c oracle/*.so

.PHONY: all host device oracle examples clean
