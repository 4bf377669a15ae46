#!/bin/bash
# GPU box: the instrumented build (-DRT_CYCLE_PROF, tools/ab_build.sh prof ...) on the given bench arguments: cycle split and lane census.
cd ${GRAFT_REPO_ROOT:-.}
RT_AMD_LIB=$PWD/ray-tracer_amd/lib/variants/librt_amd_prof.so timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 --warmup 0 "$@" 2>&1 >/dev/null | grep "rt prof"
