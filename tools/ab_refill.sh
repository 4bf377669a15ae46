#!/bin/bash
# GPU box: every variant library x {lockstep, refill} fused kernel; keeps the "[rt prof]" lines of instrumented builds.
cd "$(dirname "$0")/.."
for lib in ray-tracer_amd/lib/librt_amd.so ray-tracer_amd/lib/variants/librt_amd_*.so; do
  n=$(basename $lib .so); n=${n#librt_amd}
  for v in ${REFILL:-1}; do
    out=$(RT_MEGA_PATHS=$v RT_AMD_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 5 --warmup 1 --mode megakernel "$@" 2>gpurun_out/err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])") || exit 1
    echo "lib${n:-_default} refill=$v: Mrays/s, ms = $out"
    grep -h "rt prof" gpurun_out/err.txt | tail -1
  done
done
