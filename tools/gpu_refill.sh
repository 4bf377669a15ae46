#!/bin/bash
# GPU box: parity tests with the refill kernel, then bench lockstep vs refill.
cd ${GRAFT_REPO_ROOT:-.}
RT_MEGA_PATHS=1 timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests_refill.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests_refill.log
[ $rc -ne 0 ] && exit $rc
for v in 0 1; do
  RT_MEGA_PATHS=$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 5 --warmup 1 --mode megakernel "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('refill', $v, d['value'], 'Mrays/s', d['ms_per_step'], 'ms')" || exit 1
done
