#!/bin/bash
# GPU box: parity tests, then wide (four-child) shortcut nodes against the two-child form (RT_SHORT_BINARY=1) on the five workloads.
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3_gpu_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r3_gpu_tests.log
[ $rc -ne 0 ] && exit $rc
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']['device_requests']; print(d['value'], 'Mrays/s', d['ms_per_step'], 'ms; items', r['inner_items_visited'], 'lds', r['of_those_from_lds'], 'wide', r.get('wide_visits'), 'tri', r['triangle_tests_run'], 'rewalks', r['shortcut_rewalks'], 'rays', d['config']['rays_per_step'])"; }
for a in "--config c2" "--config c3 --spp 32" "--config c4 --spp 4" "--config c5 --spp 256" "--config c3is --spp 32"; do
  for env in "" "RT_SHORT_BINARY=1" ${EXTRA_ENVS}; do
    echo "== $a  [$env]"
    env $env timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --warmup 1 $a 2>/dev/null | line || exit 1
  done
done
