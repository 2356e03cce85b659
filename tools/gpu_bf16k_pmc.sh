#!/bin/bash
# round 4: SQ stall / LDS / MFMA counters + clock of the role-split 256 x 128 bf16x3 loop (variant 1, the default), its staging-only (11) and math-only (12)
# forms and the round-3 eight-wave loop (8) on the two judged shapes (one PMC pass per run, no tracing)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out/pmc_bf16k
for v in ${VARIANTS:-8 1 11 12}; do
for shape in ${SHAPES:-"40960 512 512 0 0" "40960 1536 512 0 0"}; do
  tag=v${v}_$(echo $shape | tr ' ' '_')
  MANSY_BF16_VARIANT=$v MANSY_FORCE_TILE=256 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d gpurun_out/pmc_bf16k/$tag -- python3 tools/gemm_pmc.py $shape 0 bf16x3 1 > gpurun_out/pmc_bf16k_$tag.log 2>&1; echo "rc=$? $tag"
  f=$(find gpurun_out/pmc_bf16k/$tag -name "*counter_collection.csv" | head -1)
  kt=$(find gpurun_out/pmc_bf16k/$tag -name "*kernel_trace.csv" | head -1)
  python3 - "$f" "$kt" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'gemm_bf16' in r['Kernel_Name']]
print(' kernel', sorted({r['Kernel_Name'][:70] for r in rows}))
agg = collections.defaultdict(float); n = collections.Counter()
for r in rows: agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
per = {k: agg[k] / n[k] for k in agg}
for k in per: print(f'  {k:28s} {per[k]:16.0f} per launch ({n[k]} launches)')
w = per['SQ_WAVE_CYCLES']
for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS'):
    print(f'  {k} / WAVE_CYCLES = {per[k]/w:.3f}')
print(f"  LDS_BANK_CONFLICT / LDS_IDX_ACTIVE = {per['SQ_LDS_BANK_CONFLICT']/max(per['SQ_LDS_IDX_ACTIVE'],1):.3f}")
try:
    d = [ (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in csv.DictReader(open(sys.argv[2])) if 'gemm_bf16' in r['Kernel_Name']]
    us = sum(d) / len(d) / 1e3
    clk = per['GRBM_GUI_ACTIVE'] / 8 / (us * 1e3)          # GHz: the counter sums the 8 XCDs
    busy = per['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (us * 1e3 * clk)   # per SIMD: MFMA cycles / elapsed cycles
    print(f"  duration {us:.1f} us (profiled), effective clock {clk:.2f} GHz, matrix-pipe busy (MFMA cycles per SIMD / elapsed cycles) = {busy:.3f}")
except Exception as e:
    print('  (no kernel trace:', e, ')')
PY
done; done
