#!/bin/bash
# GPU box: frame time and what the walk visits (bench.py's device_requests), default library and variants.  usage: tools/r3_visits.sh [configs...]
cd ${GRAFT_REPO_ROOT:-.}
for cfg in ${@:-c2 c3 c4 c5 c3is}; do
  case $cfg in c2) a="--config c2";; c3) a="--config c3 --spp 32";; c4) a="--config c4 --spp 4";; c5) a="--config c5 --spp 256";; c3is) a="--config c3is --spp 32";; esac
  for lib in ray-tracer_amd/lib/librt_amd.so ray-tracer_amd/lib/variants/librt_amd_*.so; do
    [ -e "$lib" ] || continue
    RT_AMD_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 5 --warmup 1 $a 2>gpurun_out/err.txt | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']['device_requests']; n=d['config']['rays_per_step']
print('$cfg', '$(basename $lib .so)', d['ms_per_step'], 'ms;', 'per ray: wide visits %.2f, inner items %.2f, triangle tests %.2f, rewalks %d' % (r['wide_visits']/n, r['inner_items_visited']/n, r['triangle_tests_run']/n, r['shortcut_rewalks']))" || exit 1
  done
done
