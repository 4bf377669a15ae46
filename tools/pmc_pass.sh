#!/bin/bash
# One rocprofv3 counter pass over a short bench run (on the GPU box).
#   tools/pmc_pass.sh <tag> "<counters>" [bench args...]
# Output: gpurun_out/pmc_<tag>/ (csv).  Counter runs never combine with tracing flags.
tag=$1; ctrs=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --pmc $ctrs --output-format csv -d $root/gpurun_out/pmc_$tag -- python3 $root/bench.py --no-cpu-baseline "$@" > $root/gpurun_out/pmc_$tag.log 2>&1
rc=$?; echo "pmc_$tag exit=$rc"; exit $rc
