#!/bin/bash
# GPU box: tools/round_profile.sh for the given configs, one after the other; stops at the first failure.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/profiles_new
for cfg in "$@"; do
  echo "== $cfg"
  tools/round_profile.sh $cfg r02 > gpurun_out/round_profile_$cfg.txt 2>&1 || { tail -5 gpurun_out/round_profile_$cfg.txt; exit 1; }
  tail -1 gpurun_out/round_profile_$cfg.txt | cut -c1-400
  cp profiles/r02_bench_$cfg.json profiles/r02_kernel_stats_$cfg.csv gpurun_out/profiles_new/ && cp profiles/r02_pmc_*.json gpurun_out/profiles_new/
done
