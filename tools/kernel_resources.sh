#!/bin/bash
# Register / scratch / occupancy of the render kernels' variants (compiles rt_core.hip for the device only).
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
  -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function -Iinclude "$@" --cuda-device-only -c -o /dev/null ray-tracer_amd/csrc/rt_core.hip \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|SGPRs:" | sed 's/.*remark: //' | paste - - - - - | grep -E "${FILTER:-paths_kernel|trace_kernel}" | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g'
