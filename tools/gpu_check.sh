#!/bin/bash
# GPU box: parity tests, then trace micro-benchmark at several LDS cache sizes, then bench (both modes).
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
root=$PWD
cd /tmp && export TMPDIR=/tmp
for nodes in ${LDS_SWEEP:-0 4096}; do
  RT_LDS_NODES=$nodes timeout -k 10 100 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/prof_l$nodes -- python3 $root/tools/trace_micro.py > /dev/null 2>&1 || exit 1
  echo lds_nodes $nodes
  grep -h rt_trace_kernel $root/gpurun_out/prof_l$nodes/*/*kernel_trace.csv | awk -F, '{print ($(NF-11)-$(NF-12))/1000 " us grid " $(NF-2) " lds " $(NF-10)}'
done
cd $root
for m in wavefront megakernel; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --mode $m 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['mode'], d['value'], 'Mrays/s', d['ms_per_step'], 'ms', d['roofline'])" || exit 1
done
