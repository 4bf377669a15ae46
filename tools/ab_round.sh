#!/bin/bash
# GPU box: parity tests with the default library, then tools/ab_variants.sh on the four bench scenes (c2 twice).
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py -x -q --timeout 120 --timeout-method thread > gpurun_out/ab_parity.log 2>&1
tail -2 gpurun_out/ab_parity.log
grep -q passed gpurun_out/ab_parity.log && ! grep -q failed gpurun_out/ab_parity.log || exit 1
echo == c2 && tools/ab_variants.sh --config c2 && echo == c3 && tools/ab_variants.sh --config c3 --spp 32 && echo == c4 && tools/ab_variants.sh --config c4 --spp 4 \
  && echo == c5 && tools/ab_variants.sh --config c5 --spp 256 && echo == c3is && tools/ab_variants.sh --config c3is --spp 32 && echo == c2 again && tools/ab_variants.sh --config c2
