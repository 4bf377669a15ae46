#!/bin/bash
# GPU box: sweep of the sub-tree build constants (RT_SUB_COST_INNER:RT_SUB_GAIN pairs) on a workload given as bench.py arguments.
# SWEEP="0.8:1.0 1.2:0.8" tools/gpu_sub_sweep.sh --scene scene.txt
cd ${GRAFT_REPO_ROOT:-.}
for v in ${SWEEP:-0.8:1.0}; do
  RT_SUB_COST_INNER=${v%%:*} RT_SUB_GAIN=${v##*:} timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('cost:gain $v', d['value'], 'Mrays/s', d['ms_per_step'], 'ms')" || exit 1
done
