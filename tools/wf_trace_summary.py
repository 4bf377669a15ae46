#!/usr/bin/env python3
"""Per-iteration kernel durations of the last wavefront frame in a rocprofv3 --kernel-trace csv."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 386
rows = list(csv.DictReader(open(f)))
tr = [r for r in rows if 'rt_trace_kernel<false>' in r['Kernel_Name']][-n:]
lg = [r for r in rows if 'rt_logic_kernel<false>' in r['Kernel_Name']][-n:]
d = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
td = [d(r) for r in tr]; ld = [d(r) for r in lg]
print('trace total ms %.2f logic total ms %.2f' % (sum(td) / 1e3, sum(ld) / 1e3))
edges = [0, 8, 64, 128, 192, 256, 320, n]
for a, b in zip(edges[:-1], edges[1:]):
    if b > a:
        print(a, b, 'trace avg us %.1f logic avg us %.1f' % (sum(td[a:b]) / (b - a), sum(ld[a:b]) / (b - a)))
span = (int(tr[-1]['End_Timestamp']) - int(lg[0]['Start_Timestamp'])) / 1e6
print('frame span ms %.2f gaps ms %.2f' % (span, span - (sum(td) + sum(ld)) / 1e3))
print('trace VGPR', tr[0]['VGPR_Count'], 'LDS', tr[0]['LDS_Block_Size'], 'grid', tr[0]['Grid_Size_X'], '| logic VGPR', lg[0]['VGPR_Count'], 'grid', lg[0]['Grid_Size_X'])
