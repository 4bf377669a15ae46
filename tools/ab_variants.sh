#!/bin/bash
# GPU box: bench.py with the default library and every variant built by tools/ab_build.sh, on ONE box (boxes differ by a
# few per cent); keeps the "[rt prof]" line of instrumented (-DRT_CYCLE_PROF) builds.   usage: tools/ab_variants.sh [bench args]
cd "$(dirname "$0")/.."
for lib in ray-tracer_amd/lib/librt_amd.so ray-tracer_amd/lib/variants/librt_amd_*.so; do
  [ -e "$lib" ] || continue
  n=$(basename $lib .so); n=${n#librt_amd}
  out=$(RT_AMD_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 5 --warmup 1 "$@" 2>gpurun_out/err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])") || exit 1
  echo "lib${n:-_default}: Mrays/s, ms = $out"
  grep -h "rt prof" gpurun_out/err.txt | tail -1
done
