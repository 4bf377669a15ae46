#!/bin/bash
# GPU box: frame time against the path kernel's two votes (RT_VOTE_VERTEX x RT_VOTE_LEAF), one box.   usage: tools/r3_vote_sweep.sh "<bench args>" "<vertex votes>" "<leaf votes>"
cd ${GRAFT_REPO_ROOT:-.}
a=$1
for v in $2; do
  line="vertex $v:"
  for l in $3; do
    ms=$(RT_VOTE_VERTEX=$v RT_VOTE_LEAF=$l timeout -k 10 200 python bench.py --no-cpu-baseline --steps 5 --warmup 1 $a 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])") || exit 1
    line="$line  leaf $l = $ms"
  done
  echo "$line"
done
