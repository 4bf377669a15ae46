#!/bin/bash
# GPU box: the same bench with and without one environment switch (upload-time experiments).  usage: tools/gpu_env_ab.sh VAR=value [bench args]
cd ${GRAFT_REPO_ROOT:-.}
kv=$1; shift
for on in 0 1 0 1; do
  if [ $on = 1 ]; then export "$kv"; else unset "${kv%%=*}"; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$kv' if $on else 'default', d['value'], 'Mrays/s', d['ms_per_step'], 'ms')" || exit 1
done
