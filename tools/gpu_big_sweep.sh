#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for b in ${SWEEP:-3 4 6 8 48}; do
  RT_BIG_LEAF=$b timeout -k 10 200 python bench.py --no-cpu-baseline --steps 5 --warmup 1 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('big_leaf', $b, d['value'], 'Mrays/s', d['ms_per_step'], 'ms')" || exit 1
done
