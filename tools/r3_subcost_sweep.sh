#!/bin/bash
# GPU box: frame time and steps per ray against the sub-tree builder's cost of a node visit (RT_SUB_COST_INNER, in triangle tests): smaller = shorter runs under
# the shortcut tree, more wide nodes.   usage: tools/r3_subcost_sweep.sh "<costs>" configs...
cd ${GRAFT_REPO_ROOT:-.}
costs=$1; shift
for cfg in "$@"; do
  case $cfg in c2) a="--config c2";; c3) a="--config c3 --spp 32";; c4) a="--config c4 --spp 4";; c5) a="--config c5 --spp 256";; c3is) a="--config c3is --spp 32";; esac
  for k in $costs; do
    RT_SUB_COST_INNER=$k timeout -k 10 200 python bench.py --no-cpu-baseline --steps 5 --warmup 1 $a 2>gpurun_out/err.txt | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']['device_requests']; n=d['config']['rays_per_step']
print('$cfg cost $k:', d['ms_per_step'], 'ms;', 'per ray: wide visits %.2f, triangle tests %.2f; from LDS %.3f' % (r['wide_visits']/n, r['triangle_tests_run']/n, r['of_those_from_lds']/max(r['inner_items_visited'],1)))" || exit 1
  done
done
