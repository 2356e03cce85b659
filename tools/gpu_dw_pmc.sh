#!/bin/bash
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes, no tracing) and L2 hit counters of ONE weight-gradient product
# (dW 512 x 512, K = 40 960, both operands K-major, split-K with atomics) in fp32 and bf16x3: is the split-bf16 dW re-fetching its operands?
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out/pmc_dw
for prec in f32 bf16x3; do
  for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    tag=${prec}_$(echo $ctr | tr ' ' '_')
    rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/pmc_dw/$tag -- python3 tools/gemm_pmc.py 512 512 40960 1 1 1 $prec > gpurun_out/pmc_dw_$tag.log 2>&1; echo "rc=$? $tag"
    f=$(find gpurun_out/pmc_dw/$tag -name "*counter_collection.csv" | head -1)
    python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'gemm' in r['Kernel_Name']]
agg = collections.defaultdict(float); n = collections.Counter()
for r in rows: agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in agg: print(f'  {sorted({r["Kernel_Name"][:50] for r in rows})} {k:16s} {agg[k]/n[k]:16.1f} per launch ({n[k]} launches)')
PY
  done
done
