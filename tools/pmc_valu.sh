#!/bin/bash
# GPU box: VALU instruction counts of the fused kernel for every variant library.
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root
for lib in ray-tracer_amd/lib/librt_amd.so ray-tracer_amd/lib/variants/librt_amd_*.so; do
  n=$(basename $lib .so)
  export RT_AMD_LIB=$root/$lib
  tools/pmc_pass.sh valu_$n "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" --steps 2 --warmup 0 > /dev/null || exit 1
  echo "== $n"; python3 tools/pmc_summary.py gpurun_out/pmc_valu_$n "rt_render_paths_kernel<false, false, false, false>" | sed 's/total .* per dispatch/per dispatch/'
done
