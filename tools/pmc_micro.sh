#!/bin/bash
# counter pass over tools/trace_micro.py (GPU box): tools/pmc_micro.sh <tag> "<counters>"
tag=$1; ctrs=$2
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $root/gpurun_out/pmcm_$tag -- python3 $root/tools/trace_micro.py > $root/gpurun_out/pmcm_$tag.log 2>&1
echo "pmcm_$tag exit=$?"
python3 $root/tools/pmc_summary.py $root/gpurun_out/pmcm_$tag "rt_trace_kernel<true>"
