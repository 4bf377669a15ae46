#!/bin/bash
# GPU box: tools/round_profile.sh for the given configs, one after the other; stops at the first failure.   TAG=r03 (default) names the files
cd ${GRAFT_REPO_ROOT:-.}
TAG=${TAG:-r03}
mkdir -p gpurun_out/profiles_new
for cfg in "$@"; do
  echo "== $cfg"
  tools/round_profile.sh $cfg $TAG > gpurun_out/round_profile_$cfg.txt 2>&1 || { tail -5 gpurun_out/round_profile_$cfg.txt; exit 1; }
  tail -1 gpurun_out/round_profile_$cfg.txt | cut -c1-400
  cp profiles/${TAG}_bench_$cfg.json profiles/${TAG}_kernel_stats_$cfg.csv gpurun_out/profiles_new/ && cp $(python3 -c "import json; print(json.load(open(\"profiles/${TAG}_bench_$cfg.json\"))[\"roofline\"][\"pmc\"][\"file\"])") gpurun_out/profiles_new/
done
