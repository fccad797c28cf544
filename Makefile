# Builds libeegldm.so (HIP, gfx950 only).  The oracle is pure Python (torch CPU / numpy): nothing to compile under oracle/.
PKG   := synthetic-sleep-eeg-signal-generation-using-latent-diffusion-models_amd
CSRC  := $(PKG)/csrc
HIPCC ?= /opt/rocm/bin/hipcc
# -fno-slp-vectorize -fno-vectorize: no v_pk_{fma,add,mul}_f32.  The low lane of packed-fp32 VALU instructions came out wrong in waves that
# shared a CU with the LDS-DMA weight-gradient GEMM of another stream (tools/debug/gn_hazard.sh); scalar f32 math is exact
# there, costs nothing measurable (LDM step +0.4 %, AEKL step -2.9 %), and tests/test_abi.py keeps the count of packed instructions at zero.
FLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -fno-slp-vectorize -fno-vectorize -Iinclude
SRCS  := $(wildcard $(CSRC)/*.hip)
OBJS  := $(patsubst $(CSRC)/%.hip,$(CSRC)/build/%.o,$(SRCS))
LIB   := $(PKG)/libeegldm.so

# test infrastructure (NOT linked into the product): call-recording librccl stand-in for tests/test_gpu_comm_fake.py
FAKE  := tests/fake_rccl/libfake_rccl.so

all: $(LIB) $(FAKE)

$(FAKE): tests/fake_rccl/fake_rccl.hip
	$(HIPCC) --offload-arch=gfx950 -O2 -std=c++17 -fPIC -shared $< -o $@

$(CSRC)/build/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/eegldm.h
	@mkdir -p $(CSRC)/build
	$(HIPCC) $(FLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC $(OBJS) -o $@

# developer build with per-stage cycle stamps in the GEMM kernels (tools/debug/stage_timing.py)
dbg:
	@mkdir -p $(CSRC)/build_dbg
	for f in $(SRCS); do $(HIPCC) $(FLAGS) -DEEG_STAGE_TIMING -c $$f -o $(CSRC)/build_dbg/$$(basename $$f .hip).o || exit 1; done
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC $(CSRC)/build_dbg/*.o -o tools/debug/libeegldm_dbg.so

clean:
	rm -rf $(CSRC)/build $(CSRC)/build_dbg $(LIB) $(FAKE)
.PHONY: all clean dbg
