#!/bin/bash
# GPU box: the round's evidence -- bench line, rocprofv3 kernel stats of the same command, HBM traffic counters.
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
k='rt_render_paths_kernel<false, true, false, false>'
cd $root
timeout -k 10 400 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err || exit 1
cat gpurun_out/bench_full.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_round -- python3 $root/bench.py --no-cpu-baseline > $root/gpurun_out/prof_round.log 2>&1 || exit 1
cd $root
tools/pmc_pass.sh fetch "FETCH_SIZE" --steps 2 --warmup 0 && python3 tools/pmc_summary.py gpurun_out/pmc_fetch "$k" || exit 1
tools/pmc_pass.sh write "WRITE_SIZE" --steps 2 --warmup 0 && python3 tools/pmc_summary.py gpurun_out/pmc_write "$k" || exit 1
tools/pmc_pass.sh l2 "TCC_HIT_sum TCC_MISS_sum" --steps 2 --warmup 0 && python3 tools/pmc_summary.py gpurun_out/pmc_l2 "$k"
