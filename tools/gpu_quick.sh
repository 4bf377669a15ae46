#!/bin/bash
# GPU box: parity tests then bench for the given modes/env sweeps.  usage: tools/gpu_quick.sh [bench args]
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -2 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
for nodes in ${LDS_SWEEP:-99999}; do
for m in ${MODES:-megakernel lockstep wavefront}; do
  RT_LDS_NODES=$nodes timeout -k 10 200 python bench.py --no-cpu-baseline --steps 5 --warmup 1 --mode $m "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('lds_nodes', $nodes, d['config']['mode'], d['value'], 'Mrays/s', d['ms_per_step'], 'ms', d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])" || exit 1
done; done
