#!/bin/bash
# GPU box: SQ counter passes on the fused kernel (instruction mix, lane utilisation, wait cycles).  usage: tools/pmc_sq.sh [kernel-substring]
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
k=${1:-rt_render_paths_kernel<false, false, false, false>}
cd $root
tools/pmc_pass.sh sq1 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" --steps 2 --warmup 0 && python3 tools/pmc_summary.py gpurun_out/pmc_sq1 "$k" || exit 1
tools/pmc_pass.sh sq2 "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" --steps 2 --warmup 0 && python3 tools/pmc_summary.py gpurun_out/pmc_sq2 "$k" || exit 1
tools/pmc_pass.sh sq3 "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" --steps 2 --warmup 0 && python3 tools/pmc_summary.py gpurun_out/pmc_sq3 "$k"
