#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for b in ${SWEEP:-8 16 24 32 64 96}; do
  echo "RT_BIG_LEAF=$b"; RT_BIG_LEAF=$b MODES=megakernel bash tools/gpu_c4.sh | cut -c1-60 || exit 1
done
