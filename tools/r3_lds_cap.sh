#!/bin/bash
# GPU box: frame time against the size of the LDS window (RT_LDS_NODES caps it, in items), default library and variants.
# usage: tools/r3_lds_cap.sh "<bench args>" cap...
cd ${GRAFT_REPO_ROOT:-.}
a=$1; shift
for cap in "$@"; do
  echo "== $a, window capped at $cap items"
  RT_LDS_NODES=$cap tools/ab_variants.sh $a || exit 1
done
