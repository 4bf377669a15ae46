#!/bin/bash
# GPU box: GPU tests, then lane census (instrumented build) and bench lines of the five workloads at short settings.
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3_gpu_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r3_gpu_tests.log
[ $rc -ne 0 ] && exit $rc
for a in "--config c2" "--config c3 --spp 32" "--config c4 --spp 4" "--config c5 --spp 256" "--config c3is --spp 32"; do
  echo "== $a"
  tools/gpu_prof_census.sh $a || exit 1
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --warmup 1 $a 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'Mrays/s', d['ms_per_step'], 'ms', d['roofline']['device_requests'])" || exit 1
done
