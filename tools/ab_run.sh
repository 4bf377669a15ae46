#!/bin/bash
# Run bench.py for every variant library (on the GPU box), interleaved rounds: tools/ab_run.sh [bench args...]
cd "$(dirname "$0")/.."
for round in 1 2; do
for lib in ray-tracer_amd/lib/variants/librt_amd_*.so; do
  n=$(basename $lib .so); n=${n#librt_amd_}
  for nodes in ${LDS_SWEEP:-99999}; do
  out=$(RT_LDS_NODES=$nodes RT_AMD_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 5 --warmup 1 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "round $round $n lds_nodes=$nodes: Mrays/s, ms = $out"
  done
done; done
