#!/bin/bash
# GPU box: what do the PRIMARY rays cost?  bench lines with --bounces 1 (every vertex pass ends its paths: the lanes of a wave
# take new paths together) against the full frames, default library and vote-threshold variants.  usage: tools/r3_primary_ab.sh
cd ${GRAFT_REPO_ROOT:-.}
for cfg in "--config c2" "--config c3 --spp 32" "--config c4 --spp 4" "--config c5 --spp 256" "--config c3is --spp 32"; do
  for b in 1 2 6; do
    echo "== $cfg --bounces $b"
    tools/ab_variants.sh $cfg --bounces $b || exit 1
  done
done
