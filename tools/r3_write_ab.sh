#!/bin/bash
# GPU box: HBM write bytes of the path kernel (WRITE_SIZE), default library and variants, on scene.txt.   usage: tools/r3_write_ab.sh
cd ${GRAFT_REPO_ROOT:-.}
root=$PWD
for lib in ray-tracer_amd/lib/librt_amd.so ray-tracer_amd/lib/variants/librt_amd_*.so; do
  [ -e "$lib" ] || continue
  n=$(basename $lib .so)
  RT_AMD_LIB=$root/$lib tools/pmc_pass.sh w_$n "WRITE_SIZE" --config c2 --steps 2 --warmup 0 || exit 1
  echo "== $n"; python3 tools/pmc_summary.py gpurun_out/pmc_w_$n rt_render_paths_kernel
done
