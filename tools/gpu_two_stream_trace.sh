#!/bin/bash
# Kernel trace of the VP train step WITH the two-stream decoder (two half-batches): how much of the time do kernels of the two queues really overlap?
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/ts; export TMPDIR=/tmp
cat > /tmp/vp_ts.py <<'PY'
import sys, os, random
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from mansy_immersivevideostreaming_amd.viewport_prediction.models import ViewportTransformerMTIO, FusedAdamW
from bench import synthetic_trajectories
torch.manual_seed(5); random.seed(5); np.random.seed(5)
m = ViewportTransformerMTIO(in_channel=2, fut_window=10, d_model=512, dim_feedforward=512, device='cuda').to('cuda'); m.train()
m.two_stream = bool(int(sys.argv[2]))
opt = FusedAdamW(m, lr=1e-4)
h, c, f = (t.cuda() for t in synthetic_trajectories(4096, 10, 10, seed=5))
for _ in range(int(sys.argv[1])): m.train_step(h, c, f, opt)
torch.cuda.synchronize()
PY
rm -rf gpurun_out/ts/*; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ts -- python3 /tmp/vp_ts.py 4 1 > gpurun_out/ts.log 2>&1; echo "rc=$?"
python3 - <<'PY'
import csv, glob, collections
f = sorted(glob.glob('gpurun_out/ts/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last step only: find the last adamw kernel and the one before
ad = [i for i, r in enumerate(rows) if 'adamw' in r['Kernel_Name']]
lo, hi = ad[-2] + 1, ad[-1] + 1
step = rows[lo:hi]
t0, t1 = int(step[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in step)
print('kernels', len(step), 'span ms', (t1 - t0) / 1e6, 'sum of durations ms', sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step) / 1e6)
qs = collections.Counter((r['Queue_Id'], r['Stream_Id']) for r in step)
print('queues/streams', qs)
# time with >= 2 kernels in flight / exactly 1 / 0
ev = []
for r in step:
    ev.append((int(r['Start_Timestamp']), 1)); ev.append((int(r['End_Timestamp']), -1))
ev.sort()
cur, last, acc = 0, t0, collections.Counter()
for t, d in ev:
    acc[min(cur, 3)] += t - last; last = t; cur += d
print('time by kernels in flight (ms):', {k: v / 1e6 for k, v in sorted(acc.items())})
# per queue busy
for q in qs:
    b = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step if (r['Queue_Id'], r['Stream_Id']) == q)
    print('queue', q, 'busy ms', b / 1e6)
# a window of the decoder forward: print 60 kernels from the middle of the two-queue phase
two = [i for i, r in enumerate(step) if (r['Queue_Id'], r['Stream_Id']) != (step[0]['Queue_Id'], step[0]['Stream_Id'])]
if two:
    s = two[len(two) // 8]
    for r in step[s:s + 70]:
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} q{r['Queue_Id']} s{r['Stream_Id']} {r['Kernel_Name'][:60]}")
PY
