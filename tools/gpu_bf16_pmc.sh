#!/bin/bash
# SQ stall / LDS / MFMA counters of the bf16x3 loops with pre-split weights (one PMC pass per shape, no tracing): where the waves of
# gemm_bf16f_kernel (128x128 tiles, [40 960-row] product) and gemm_bf16p_kernel (64x64 tiles, [4 096-row] decoder product) spend their cycles.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out/pmc_bf16
for shape in "40960 1536 512 0 0" "40960 512 512 0 0" "4096 512 512 0 0"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES \
    --output-format csv -d gpurun_out/pmc_bf16/$tag -- python3 tools/gemm_pmc.py $shape 0 bf16x3 1 > gpurun_out/pmc_bf16_$tag.log 2>&1; echo "rc=$? $tag"
  f=$(find gpurun_out/pmc_bf16/$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'gemm_bf16' in r['Kernel_Name']]
print(' kernel', sorted({r['Kernel_Name'][:60] for r in rows}))
agg = collections.defaultdict(float); n = collections.Counter()
for r in rows: agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in agg: print(f'  {k:28s} {agg[k]/n[k]:16.0f} per launch ({n[k]} launches)')
w = agg['SQ_WAVE_CYCLES'] / max(n['SQ_WAVE_CYCLES'], 1)
for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS'):
    print(f'  {k} / WAVE_CYCLES = {agg[k]/max(n[k],1)/w:.3f}')
print(f"  LDS_BANK_CONFLICT / LDS_IDX_ACTIVE = {agg['SQ_LDS_BANK_CONFLICT']/max(agg['SQ_LDS_IDX_ACTIVE'],1):.3f}")
print(f"  MFMA_BUSY_CYCLES / (4 x WAVE_CYCLES quad-cycles) = {agg['SQ_VALU_MFMA_BUSY_CYCLES']/max(n['SQ_VALU_MFMA_BUSY_CYCLES'],1)/(4*w):.3f}")
PY
done
