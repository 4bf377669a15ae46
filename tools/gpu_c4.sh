#!/bin/bash
# GPU box: BASELINE config 4 (synthetic 10M-triangle field, 3840x2160, 4 spp) in the given modes.
cd ${GRAFT_REPO_ROOT:-.}
for m in ${MODES:-megakernel lockstep}; do
  timeout -k 10 500 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --mode $m --synth-instances 628 --width 3840 --height 2160 --spp 4 "$@" 2>gpurun_out/c4_err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['mode'], d['value'], 'Mrays/s', d['ms_per_step'], 'ms', d['roofline'])" || exit 1
done
