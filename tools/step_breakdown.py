#!/usr/bin/env python3
"""Per-kernel (name, grid) breakdown of the last train step in a rocprofv3 kernel trace (gpurun_out/prof)."""
import csv, glob, collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', 'prof', '**', '*_kernel_trace.csv'), recursive=True), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adamw' in r['Kernel_Name']]
step = rows[idx[-2] + 1:idx[-1] + 1]
short = lambda n: n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:52]
agg = collections.OrderedDict()
for r in step:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    a = agg.setdefault((short(r['Kernel_Name']), r['Grid_Size_X']), [0, 0.0, 1e9, 0.0])
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[1]) if len(sys.argv) > 1 else 40]:
    print(f'{k[0]:52s} grid {k[1]:>9s} n={a[0]:4d} tot={a[1] / 1e3:7.3f} ms avg={a[1] / a[0]:7.1f} min={a[2]:7.1f} max={a[3]:7.1f}')
span = (int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e6
print('kernels', len(step), 'busy ms', round(sum(a[1] for a in agg.values()) / 1e3, 3), 'span ms', round(span, 3))
