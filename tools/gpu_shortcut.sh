#!/bin/bash
# GPU box: all GPU tests, then the bench workloads with the reference's walk and with the shortcut tree (ms per step).
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -s > gpurun_out/shortcut_tests.log 2>&1; rc=$?
tail -3 gpurun_out/shortcut_tests.log
grep -h "inner items per ray\|coplanar sheets" gpurun_out/shortcut_tests.log
[ $rc -ne 0 ] && exit $rc
run() {
  for w in reference shortcut; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --warmup 1 --walk $w "$@" 2>gpurun_out/bench_err.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); q=d['roofline']['device_requests']; print('  $w', d['value'], 'Mrays/s', d['ms_per_step'], 'ms; inner items / ray', round(q['inner_items_visited']/d['config']['rays_per_step'],2), 'from LDS', round(q['of_those_from_lds']/max(q['inner_items_visited'],1),3), 'tri tests / ray', round(q['triangle_tests_run']/d['config']['rays_per_step'],2), 'rewalks', q['shortcut_rewalks'])" || { tail -5 gpurun_out/bench_err.log; exit 1; }
  done
}
[ -n "$FUZZ" ] && { timeout -k 10 600 python tools/gpu_fuzz.py $FUZZ 7000 && timeout -k 10 600 python tools/gpu_fuzz_render.py ${FUZZ_RENDER:-200} 9000 || exit 1; }
[ -n "$NOBENCH" ] && exit 0
echo == c2 && run --config c2 && echo == c3 32spp && run --config c3 --spp 32 && echo == c4 4spp && run --config c4 --spp 4 && echo == c3is 32spp && run --config c3is --spp 32 && echo == c5 256spp && run --config c5 --spp 256
