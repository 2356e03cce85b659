#!/bin/bash
# kernel durations (rocprofv3 trace) of the [4096, 512] x K product for K = 32 .. 1024: the device-side fixed cost of a launch
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out/prof_ovh
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ovh -- python3 tools/gemm_overhead.py > gpurun_out/prof_ovh.log 2>&1; echo "rc=$?"
f=$(find gpurun_out/prof_ovh -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'gemm_f32' in r['Kernel_Name']]
# launches come in groups of 210 per (K, epilogue) in program order
dur = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in sorted(rows, key=lambda r: int(r['Start_Timestamp']))]
n = 210
for i, K in enumerate((32, 64, 128, 256, 512, 1024)):
    a = dur[(2 * i) * n:(2 * i + 1) * n]; b = dur[(2 * i + 1) * n:(2 * i + 2) * n]
    if a and b:
        print(f'K={K:5d}: plain {sorted(a)[len(a)//2]/1e3:6.2f} us   bias+resid {sorted(b)[len(b)//2]/1e3:6.2f} us (median kernel duration)')
PY
