#!/bin/bash
# GPU box: cycle split of the path kernel WITHOUT counters (variant prof2 = -DRT_CYCLE_PROF=2), then bench A/B of every variant library.
cd ${GRAFT_REPO_ROOT:-.}
for a in "--config c2" "--config c3 --spp 32" "--config c4 --spp 4" "--config c5 --spp 256"; do
  echo "== $a"
  RT_AMD_LIB=$PWD/ray-tracer_amd/lib/variants/librt_amd_prof2.so timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 --warmup 0 $a 2>&1 >/dev/null | grep "rt prof2" | tail -1
  [ -n "$BINARY_TOO" ] && RT_SHORT_BINARY=1 RT_AMD_LIB=$PWD/ray-tracer_amd/lib/variants/librt_amd_prof2.so timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 --warmup 0 $a 2>&1 >/dev/null | grep "rt prof2" | tail -1
  tools/ab_variants.sh $a || exit 1
done
