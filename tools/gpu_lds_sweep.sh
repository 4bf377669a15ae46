#!/bin/bash
# GPU box: frame time against the size of the LDS node window (RT_LDS_NODES caps it).  usage: tools/gpu_lds_sweep.sh "<bench args>" n1 n2 ...
cd ${GRAFT_REPO_ROOT:-.}
args=$1; shift
for n in "$@"; do
  RT_LDS_NODES=$n timeout -k 10 200 python bench.py --no-cpu-baseline --steps 5 --warmup 1 $args 2>gpurun_out/err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); q=d['roofline']['device_requests']; print('  lds nodes <= $n:', d['ms_per_step'], 'ms; from LDS', round(q['of_those_from_lds']/max(q['inner_items_visited'],1),3))" || exit 1
done
