#!/bin/bash
# GPU box: bench with and without the block-level ray pool (RT_POOL=1) on the bench scenes; extra env (RT_POOL_*) passes through.
cd ${GRAFT_REPO_ROOT:-.}
run() { out=$(timeout -k 10 200 python bench.py --no-cpu-baseline --steps 5 --warmup 1 "$@" 2>gpurun_out/err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])") || { tail -3 gpurun_out/err.txt; return 1; }; echo "$out"; }
# CONFIGS="c2;c3 --spp 32" overrides the list (semicolon separated bench arguments)
IFS=';' read -ra LIST <<< "${CONFIGS:-c2;c3 --spp 32;c4 --spp 4;c5 --spp 256;c3is --spp 32;c2}"
for cfg in "${LIST[@]}"; do
  echo "== $cfg"
  echo -n "paths kernel: "; RT_POOL=0 run --config $cfg || exit 1
  echo -n "pool kernel:  "; RT_POOL=1 run --config $cfg || exit 1
done
