#!/bin/bash
# GPU box: the round's evidence for ONE workload -- counter passes, rocprofv3 kernel stats of the bench command, then
# the bench line that reads them.    usage: tools/round_profile.sh <config c2|c3|c3is|c4|c5> [tag, default r03] [bench args]
# Writes profiles/<tag>_pmc_<workload>.json, profiles/<tag>_kernel_stats_<config>.csv, profiles/<tag>_bench_<config>.json.
cfg=${1:-c2}; tag=${2:-r03}; shift 2 2>/dev/null
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root
A="--config $cfg --steps 2 --warmup 0 $*"
tools/pmc_pass.sh ${cfg}_sq "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" $A || exit 1
tools/pmc_pass.sh ${cfg}_fetch "FETCH_SIZE" $A || exit 1
tools/pmc_pass.sh ${cfg}_write "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" $A || exit 1
tools/pmc_pass.sh ${cfg}_tatd "TA_TA_BUSY_sum TD_TD_BUSY_sum SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" $A || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$cfg -- python3 $root/bench.py --config $cfg --no-cpu-baseline "$@" > $root/gpurun_out/prof_$cfg.log 2>&1 || exit 1
cd $root
python3 tools/pmc_to_json.py $tag $cfg "$@" || exit 1
timeout -k 10 600 python3 bench.py --config $cfg "$@" > profiles/${tag}_bench_$cfg.json 2> gpurun_out/bench_$cfg.err || exit 1
cat profiles/${tag}_bench_$cfg.json
