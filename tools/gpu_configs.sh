#!/bin/bash
# GPU box: BASELINE configs 3, 4 and 5 on one GPU (parity cases measured for DESIGN.md, not the bench line).
cd ${GRAFT_REPO_ROOT:-.}
run() { timeout -k 10 500 python bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" 2>gpurun_out/cfg_err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['config']['mode'], d['config']['spp_per_gpu'], 'spp', d['value'], 'Mrays/s', d['ms_per_step'], 'ms', 'frac', d['roofline']['frac'], 'B/ray', d['roofline']['bytes_per_ray'])" || exit 1; }
run --scene scene2.txt --spp 256 && run --scene scene2_is.txt --spp 256 && run --scene scene_cornellbox.txt --spp 4096 &&
run --synth-instances 628 --width 3840 --height 2160 --spp 4 && run --synth-instances 628 --width 3840 --height 2160 --spp 128 &&
run --synth-instances 628 --width 3840 --height 2160 --spp 4 --mode lockstep && run --mode lockstep && run --mode wavefront
