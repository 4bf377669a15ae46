#!/bin/bash
# GPU box: config 4 with the shortcut tree over leaves (default there) and over runs: load stages and frame time.
cd ${GRAFT_REPO_ROOT:-.}
for m in 1048576 100000000; do
  echo "== RT_SHORT_RUNS_MAX=$m"
  RT_SHORT_RUNS_MAX=$m RT_UPLOAD_TIMING=1 timeout -k 10 300 python tools/c4_load.py 2>&1 | grep -E "triangles:|shortcut|sub-tree build|total|upload" || exit 1
  RT_SHORT_RUNS_MAX=$m timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --warmup 1 --config c4 --spp 4 2>gpurun_out/bench_err.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); q=d['roofline']['device_requests']; print('  ', d['value'], 'Mrays/s', d['ms_per_step'], 'ms; inner items / ray', round(q['inner_items_visited']/d['config']['rays_per_step'],2), 'from LDS', round(q['of_those_from_lds']/max(q['inner_items_visited'],1),3), 'tri tests / ray', round(q['triangle_tests_run']/d['config']['rays_per_step'],2), 'rewalks', q['shortcut_rewalks'])" || exit 1
done
