#!/bin/bash
# GPU box: sweep of the sub-tree thresholds (RT_BIG_LEAF:RT_SUB_COST_INNER pairs) on a workload given as bench.py arguments.
# SWEEP="16:4 8:4" tools/gpu_sub_sweep.sh --scene scene.txt
cd ${GRAFT_REPO_ROOT:-.}
for v in ${SWEEP:-16:4}; do
  RT_BIG_LEAF=${v%%:*} RT_SUB_COST_INNER=${v##*:} timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('big:sub $v', d['value'], 'Mrays/s', d['ms_per_step'], 'ms')" || exit 1
done
