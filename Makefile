# Build everything in-tree (built libraries are git-ignored but travel to the GPU box).
#   make            -> host scene lib, HIP core, oracle
#   make oracle     -> oracle only (test infrastructure)
HIPCC   ?= /opt/rocm/bin/hipcc
CXX     ?= g++
PKG     := ray-tracer_amd
LIBDIR  := $(PKG)/lib
CXXFLAGS := -std=c++17 -O2 -fPIC -pthread -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Iinclude
# -ffp-contract=off: the closest-hit arithmetic must be evaluated exactly as written
# (no FMA) so primitive ids match the oracle bit for bit.
# -fno-slp-vectorize: packed v_pk_*_f32 pairs cost register shuffles and issue no faster here (measured
# 39.2 -> 36.2 ms per frame on the bench scene with everything scalar, profiles/r01_ab_megakernel.txt).
HIPFLAGS := -std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
            -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -Iinclude

all: $(LIBDIR)/librt_scene.so $(LIBDIR)/librt_amd.so $(LIBDIR)/rtcli oracle

$(LIBDIR)/librt_scene.so: $(PKG)/csrc/host/rt_scene.cpp include/rt_scene.h include/rt.h
	@mkdir -p $(LIBDIR)
	$(CXX) $(CXXFLAGS) -shared -o $@ $<

$(LIBDIR)/librt_amd.so: $(PKG)/csrc/rt_core.hip $(wildcard $(PKG)/csrc/*.h) include/rt.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $< -ldl

$(LIBDIR)/rtcli: $(PKG)/csrc/cli/rtcli.cpp $(LIBDIR)/librt_scene.so $(LIBDIR)/librt_amd.so include/rt.h include/rt_scene.h
	$(CXX) $(CXXFLAGS) -o $@ $< -L$(LIBDIR) -lrt_scene -lrt_amd -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib

oracle: oracle/lib/librt_oracle.so
oracle/lib/librt_oracle.so: oracle/rt_oracle.cpp oracle/rt_pmath_oracle.h include/rt.h
	$(MAKE) -C oracle

clean:
	rm -rf $(LIBDIR) oracle/lib

.PHONY: all oracle clean
