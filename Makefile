# Builds the gfx950 C-ABI library (flope_amd/lib/libflope_amd.so) and the CPU-only
# host test harness (tests/host_harness/libflope_host_harness.so).
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
CSRC     := flope_amd/csrc
OBJDIR   := build/obj
LIB      := flope_amd/lib/libflope_amd.so
HARNESS  := tests/host_harness/libflope_host_harness.so
SRCS     := $(wildcard $(CSRC)/*.hip)
OBJS     := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(SRCS))
HDRS     := $(wildcard $(CSRC)/*.h) include/flope_amd.h
HIPFLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Wno-unused-value -Iinclude

all: $(LIB) $(HARNESS)

$(OBJDIR)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p $(dir $@)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

$(HARNESS): tests/host_harness/harness.cpp $(CSRC)/pose_math.h $(CSRC)/host_pack.h $(CSRC)/w4_sched.h
	g++ -O2 -fPIC -shared -std=c++17 -I$(CSRC) -o $@ $<

# the same host code under AddressSanitizer + UBSan (sanitizers run on the CPU build only):
#   make asan-test      (builds the sanitized harness and runs tests/test_host.py against it; leak detection is off because the
#                        interpreter's own libcrypto allocations are reported as leaks at exit)
HARNESS_ASAN := tests/host_harness/libflope_host_harness_asan.so
$(HARNESS_ASAN): tests/host_harness/harness.cpp $(CSRC)/pose_math.h $(CSRC)/host_pack.h $(CSRC)/w4_sched.h
	g++ -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -fPIC -shared -std=c++17 -I$(CSRC) -o $@ $<
asan: $(HARNESS_ASAN)
asan-test: $(HARNESS_ASAN)
	ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$$(gcc -print-file-name=libasan.so) FLOPE_HOST_HARNESS=$(HARNESS_ASAN) python -m pytest tests/test_host.py -q

# diagnostic build (ablation bits + in-kernel clock stamps of conv_stag; results of dbg options are wrong by construction):
#   make dbg && FLOPE_AMD_LIB=build/dbg/libflope_amd_dbg.so python tools/clock_probe.py
DBGDIR   := build/dbg
DBGOBJS  := $(patsubst $(CSRC)/%.hip,$(DBGDIR)/%.o,$(SRCS))
$(DBGDIR)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(DBGDIR)
	$(HIPCC) $(HIPFLAGS) -DFLOPE_STAG_DBG -c $< -o $@
dbg: $(DBGOBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $(DBGDIR)/libflope_amd_dbg.so $(DBGOBJS)

clean:
	rm -rf build $(LIB) $(HARNESS)

# stand-alone measurement programs used by tools/collect_profiles.sh and DESIGN.md section 9 (not part of the library)
TOOLBINS := build/fetch_calib build/launch_floor build/loop_probe build/loop_probe32 build/dma_issue_probe
build/fetch_calib: tools/calib/fetch_calib.hip
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -o $@ $<
build/launch_floor: tools/calib/launch_floor.hip
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -o $@ $<
build/loop_probe: tools/probes/loop_probe.hip
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -o $@ $<
build/loop_probe32: tools/probes/loop_probe32.hip
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -o $@ $<
build/dma_issue_probe: tools/probes/dma_issue_probe.hip
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -o $@ $<
tools: $(TOOLBINS)

.PHONY: all clean dbg tools asan asan-test
