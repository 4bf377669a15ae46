#!/bin/bash
# GPU box: config 4 (10M-triangle field, 4K, 4 spp) for A/B libraries and big-leaf thresholds: tools/gpu_c4_ab.sh "lib:threshold" ...
cd ${GRAFT_REPO_ROOT:-.}
for v in "$@"; do
  lib=${v%%:*}; b=${v##*:}
  if [ "$lib" != default ]; then export RT_AMD_LIB=ray-tracer_amd/lib/variants/librt_amd_$lib.so; else unset RT_AMD_LIB; fi
  RT_BIG_LEAF=$b timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --synth-instances 628 --width 3840 --height 2160 --spp 4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], 'Mrays/s', d['ms_per_step'], 'ms')" || exit 1
done
