#!/bin/bash
# SQ counters of the row-owning chain lab (chain of 8), two PMC passes, no tracing.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out/pmc_rowown
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_rowown/p$i -- tools/_bin/rowown_lab 8 > gpurun_out/pmc_rowown_$i.log 2>&1; echo "rc=$? pass $i"
  f=$(find gpurun_out/pmc_rowown/p$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'rowown_chain' in r['Kernel_Name']]
agg = collections.defaultdict(float); n = collections.Counter()
for r in rows: agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
w = agg['SQ_WAVE_CYCLES'] / max(n['SQ_WAVE_CYCLES'], 1)
for k in agg: print(f'  {k:30s} {agg[k]/n[k]:16.0f} per launch ({n[k]} launches)  / WAVE_CYCLES = {agg[k]/n[k]/w:.3f}')
PY
done
