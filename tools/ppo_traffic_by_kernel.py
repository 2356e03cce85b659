#!/usr/bin/env python3
"""HBM-side traffic of the PPO cycle BY KERNEL, from the four PMC passes of tools/gpu_prof_ppo.sh (FETCH_SIZE / WRITE_SIZE at 2 and at 10 timed cycles;
per cycle = the difference / (8 x bench.PPO_BLOCKS: every run times PPO_BLOCKS blocks of its cycle count); FETCH_SIZE x2 per the gfx950 correction).  python3 tools/ppo_traffic_by_kernel.py > profiles/rNN_ppo_traffic_by_kernel.txt"""
import csv, glob, os, sys, collections
sys.path.insert(0, os.getcwd())
import bench
DN = 8.0 * bench.PPO_BLOCKS


def latest(c, n):
    return sorted(glob.glob(f'gpurun_out/prof_ppo_{c}_{n}/**/*counter_collection.csv', recursive=True), key=os.path.getmtime)[-1]


def by_kernel(c, n):
    d = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(latest(c, n))):
        if r['Counter_Name'] == c:
            d[r['Kernel_Name']][0] += float(r['Counter_Value']) * 1024.0
            d[r['Kernel_Name']][1] += 1
    return d


out = {}
for c, mul in (('FETCH_SIZE', 2.0), ('WRITE_SIZE', 1.0)):
    a, b = by_kernel(c, 10), by_kernel(c, 2)
    for k in a:
        out.setdefault(k, {})[c] = mul * (a[k][0] - b.get(k, [0, 0])[0]) / DN
        out[k]['calls'] = (a[k][1] - b.get(k, [0, 0])[1]) / DN
tot = sum(v.get('FETCH_SIZE', 0) + v.get('WRITE_SIZE', 0) for v in out.values())
print(f'PPO cycle (256 envs x 16 steps): {tot / 1e6:.1f} MB per cycle at the L2 <-> fabric boundary = {tot / 4096 / 1e3:.1f} KB per env-step')
print(f'{"kernel":92s} {"calls":>6s} {"fetch MB":>9s} {"write MB":>9s} {"MB/call":>8s} {"share":>6s}')
for k, v in sorted(out.items(), key=lambda kv: -(kv[1].get('FETCH_SIZE', 0) + kv[1].get('WRITE_SIZE', 0))):
    f, w = v.get('FETCH_SIZE', 0), v.get('WRITE_SIZE', 0)
    if v['calls'] < 0.5 or f + w < 1e5:
        continue
    print(f'{k[:92]:92s} {v["calls"]:6.1f} {f / 1e6:9.1f} {w / 1e6:9.1f} {(f + w) / v["calls"] / 1e6:8.2f} {(f + w) / tot * 100:5.1f}%')
