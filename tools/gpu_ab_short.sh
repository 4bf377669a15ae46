#!/bin/bash
# GPU box: default library and the variants of tools/ab_build.sh on the shortcut walk: c2, c3 32 spp, c4 4 spp, c3is 32 spp.
cd ${GRAFT_REPO_ROOT:-.}
echo == c2 && tools/ab_variants.sh --config c2 && echo == c3 && tools/ab_variants.sh --config c3 --spp 32 && echo == c4 && tools/ab_variants.sh --config c4 --spp 4 \
  && echo == c3is && tools/ab_variants.sh --config c3is --spp 32 && echo == c2 again && tools/ab_variants.sh --config c2
