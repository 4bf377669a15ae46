#!/bin/bash
# GPU box: counters of the fused kernel on BASELINE config 4 (synthetic 10M-triangle field).
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
k=${1:-rt_render_paths_kernel<false, true, false, false>}
A="--steps 1 --warmup 0 --synth-instances 628 --width 3840 --height 2160 --spp 4"
cd $root
tools/pmc_pass.sh c4a "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE TA_TA_BUSY_sum TD_TD_BUSY_sum" $A && python3 tools/pmc_summary.py gpurun_out/pmc_c4a "$k" || exit 1
tools/pmc_pass.sh c4b "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" $A && python3 tools/pmc_summary.py gpurun_out/pmc_c4b "$k" || exit 1
tools/pmc_pass.sh c4c "FETCH_SIZE" $A && python3 tools/pmc_summary.py gpurun_out/pmc_c4c "$k"
