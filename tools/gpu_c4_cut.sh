#!/bin/bash
# GPU box: config 4 (4 spp frames) with the shortcut tree's leaves 0 .. 6 levels below the reference's leaves; load stages for the default.
cd ${GRAFT_REPO_ROOT:-.}
RT_UPLOAD_TIMING=1 timeout -k 10 300 python tools/c4_load.py 2>&1 | grep -E "triangles:|shortcut|sub-tree" || exit 1
for cut in default 0 2 4 6; do
  [ $cut = default ] && unset RT_SHORT_CUT || export RT_SHORT_CUT=$cut
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --warmup 1 --config c4 --spp 4 2>gpurun_out/bench_err.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); q=d['roofline']['device_requests']; print('  cut $cut:', d['value'], 'Mrays/s', d['ms_per_step'], 'ms; inner items / ray', round(q['inner_items_visited']/d['config']['rays_per_step'],2), 'from LDS', round(q['of_those_from_lds']/max(q['inner_items_visited'],1),3), 'tri tests / ray', round(q['triangle_tests_run']/d['config']['rays_per_step'],2), 'rewalks', q['shortcut_rewalks'])" || exit 1
done
