#!/bin/bash
# GPU box: the core parity tests, then one bench line per workload (and per variant library, if any).  usage: tools/r3_quick.sh [configs...]
cd ${GRAFT_REPO_ROOT:-.}
if [ -z "$NO_TESTS" ]; then
  timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r3_quick_tests.log 2>&1; rc=$?
  tail -3 gpurun_out/r3_quick_tests.log
  [ $rc -ne 0 ] && exit $rc
fi
for cfg in ${@:-c2 c3 c4 c5 c3is}; do
  case $cfg in c2) a="--config c2";; c3) a="--config c3 --spp 32";; c4) a="--config c4 --spp 4";; c5) a="--config c5 --spp 256";; c3is) a="--config c3is --spp 32";; esac
  echo "== $a"
  tools/ab_variants.sh $a || exit 1
done
