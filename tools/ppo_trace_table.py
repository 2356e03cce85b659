"""Per-(kernel, grid) time table of a rocprofv3 kernel trace of the PPO cycle:  python tools/ppo_trace_table.py <kernel_trace.csv> [cycles]"""
import csv, collections, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 14
by = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    n = re.sub(r'\(.*', '', n)[:40]
    by[(n, r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
out = sorted(((sum(v) / NC, k, len(v) / NC, sum(v) / len(v), min(v), max(v)) for k, v in by.items()), reverse=True)
for t, k, n, avg, mn, mx in out[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{k[0]:40s} grid {k[1]:>8s},{k[2]:>3s},{k[3]:>3s} n/cyc {n:5.1f} avg {avg:6.1f} min {mn:5.1f} max {mx:6.1f}  us/cyc {t:7.1f}")
print('total us/cyc', round(sum(o[0] for o in out), 1), 'launches/cyc', round(sum(o[2] for o in out), 1))
