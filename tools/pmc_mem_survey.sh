#!/bin/bash
# GPU box: where the memory path of rt_render_paths_kernel spends its time -- four counter passes (address translation,
# L1 / L2 latency and stalls) over a short bench run.   usage: tools/pmc_mem_survey.sh <tag> [bench args]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $root
A="--steps 2 --warmup 0 $*"
tools/pmc_pass.sh ${tag}_mA "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE" $A || exit 1
tools/pmc_pass.sh ${tag}_mB "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_LATENCY_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_BUSY_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" $A || exit 1
tools/pmc_pass.sh ${tag}_mC "TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_REQ_sum TCC_READ_sum" $A || exit 1
tools/pmc_pass.sh ${tag}_mD "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum" $A || exit 1
for p in mA mB mC mD; do python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_$p rt_render_paths_kernel; done > gpurun_out/mem_survey_$tag.txt
cat gpurun_out/mem_survey_$tag.txt
