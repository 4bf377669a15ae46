#!/bin/bash
# GPU box: RT_POOL=1 bench over a list of environment settings.  usage: tools/pool_sweep.sh "<bench args>" "ENV=.. ENV=.." "ENV=.." ...
cd ${GRAFT_REPO_ROOT:-.}
args=$1; shift
for envs in "$@"; do
  out=$(env RT_POOL=1 $envs timeout -k 10 200 python bench.py --no-cpu-baseline --steps 4 --warmup 1 $args 2>gpurun_out/err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])") || { tail -3 gpurun_out/err.txt; exit 1; }
  echo "$args [$envs]: $out"
done
