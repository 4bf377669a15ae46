#!/bin/bash
# Run bench.py for every variant library (on the GPU box): tools/ab_run.sh [bench args...]
cd "$(dirname "$0")/.."
for lib in ray-tracer_amd/lib/variants/librt_amd_*.so; do
  n=$(basename $lib .so); n=${n#librt_amd_}
  out=$(RT_AMD_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])")
  echo "$n: Mrays/s,kernel_ms = $out"
done
