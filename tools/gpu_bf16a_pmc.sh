#!/bin/bash
# SQ stall / LDS / MFMA counters of the bf16-storage products (round 6; one PMC pass per shape, no tracing) -> gpurun_out/r06_bf16_sq_pmc.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out/pmc_bf16a
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
for shape in "40960 512 512 nn" "40960 1536 512 nn" "40960 512 1536 nn" "4096 512 512 nn" "512 512 40960 tn" "1536 512 40960 tn"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d gpurun_out/pmc_bf16a/$tag -- python3 tools/gemm_bf16a_pmc.py $shape > gpurun_out/pmc_bf16a_$tag.log 2>&1; echo "rc=$? [$shape]"
  f=$(find gpurun_out/pmc_bf16a/$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'gemm_bf16a' in r['Kernel_Name']]
print(' kernel', sorted({r['Kernel_Name'][:80] for r in rows}))
agg = collections.defaultdict(float); n = collections.Counter()
for r in rows: agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in sorted(agg): print(f'  {k:28s} {agg[k]/n[k]:16.0f} per launch ({n[k]} launches)')
w = agg['SQ_WAVE_CYCLES'] / max(n['SQ_WAVE_CYCLES'], 1)
for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS'):
    print(f'  {k} / WAVE_CYCLES = {agg[k]/max(n[k],1)/w:.3f}')
print(f"  LDS_BANK_CONFLICT / LDS_IDX_ACTIVE = {agg['SQ_LDS_BANK_CONFLICT']/max(agg['SQ_LDS_IDX_ACTIVE'],1):.3f}")
gui = agg['GRBM_GUI_ACTIVE'] / max(n['GRBM_GUI_ACTIVE'], 1)
mf = agg['SQ_VALU_MFMA_BUSY_CYCLES'] / max(n['SQ_VALU_MFMA_BUSY_CYCLES'], 1)
# matrix-pipe busy as profiles/r04_bf16k_sq_pmc.txt defines it: MFMA busy cycles per SIMD (1024 SIMDs) over the launch's elapsed cycles (GRBM_GUI_ACTIVE sums the 8 XCDs)
el = gui / 8.0
print(f"  matrix-pipe busy (MFMA busy cycles per SIMD / elapsed cycles) = {mf / 1024.0 / max(el, 1):.3f}   (elapsed {el:.0f} cycles per XCD)")
PY
done
