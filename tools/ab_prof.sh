#!/bin/bash
# For every variant library: kernel-trace one frame and print the per-iteration summary (GPU box).
cd "$(dirname "$0")/.."
for lib in ray-tracer_amd/lib/variants/librt_amd_*.so; do
  n=$(basename $lib .so); n=${n#librt_amd_}
  export RT_AMD_LIB=$PWD/$lib
  tools/prof_trace.sh ab_$n "$@" > /dev/null
  echo "== $n"; python3 tools/wf_trace_summary.py gpurun_out/prof_ab_$n | head -4
done
