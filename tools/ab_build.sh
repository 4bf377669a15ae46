#!/bin/bash
# Build A/B variants of the HIP core: tools/ab_build.sh name "-DFLAG=..." [name2 "flags2" ...]
# -> ray-tracer_amd/lib/variants/librt_amd_<name>.so  (select with RT_AMD_LIB=<path>)
set -e
cd "$(dirname "$0")/.."
mkdir -p ray-tracer_amd/lib/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
    -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function -Iinclude $flags -shared \
    -o ray-tracer_amd/lib/variants/librt_amd_$name.so ray-tracer_amd/csrc/rt_core.hip \
    -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A9 "rt_render_kernelILb0" | grep -E "VGPRs:|Scratch|Occupancy" | sed "s/.*remark: */  $name:/" &
done
wait
