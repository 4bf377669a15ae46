#!/bin/bash
# GPU box: counters of the fused kernel on scene2.txt 1080p (BASELINE config 3, 16 spp sample).
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
k='rt_render_paths_kernel<false, true, false, false>'
A="--steps 1 --warmup 0 --scene scene2.txt --spp 16"
cd $root
tools/pmc_pass.sh c3a "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE TA_TA_BUSY_sum TD_TD_BUSY_sum" $A && python3 tools/pmc_summary.py gpurun_out/pmc_c3a "$k" || exit 1
tools/pmc_pass.sh c3b "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" $A && python3 tools/pmc_summary.py gpurun_out/pmc_c3b "$k"
