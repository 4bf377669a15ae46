#!/bin/bash
# GPU box: the non-parity SAH tree against the reference tree: bench scene and the 10M-triangle field.
cd ${GRAFT_REPO_ROOT:-.}
run() { timeout -k 10 500 python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>gpurun_out/sah_err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], '|', d['value'], 'Mrays/s', d['ms_per_step'], 'ms', 'B/ray', d['roofline']['bytes_per_ray'])" || { tail -5 gpurun_out/sah_err.txt; exit 1; }; }
run && run --bvh sah && run --scene scene2.txt --spp 64 && run --scene scene2.txt --spp 64 --bvh sah && run --synth-instances 628 --width 3840 --height 2160 --spp 4 --bvh sah
