#!/bin/bash
# rocprofv3 kernel trace of one bench frame (GPU box): tools/prof_trace.sh <tag> [bench args]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $root/gpurun_out/prof_$tag.log 2>&1
echo "prof_$tag exit=$?"
