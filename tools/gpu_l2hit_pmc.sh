#!/bin/bash
# L2 hit rate of the big forward products (TCC_HIT_sum / TCC_MISS_sum, one PMC pass per shape and mode, no tracing)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out/pmc_l2
for shape in "40960 1536 512 0 0" "40960 512 512 0 0" "40960 512 1536 0 1"; do
  for prec in f32 bf16x3; do
    tag=$(echo $shape | tr ' ' '_')_$prec
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_l2/$tag -- python3 tools/gemm_pmc.py $shape 0 $prec 1 > gpurun_out/pmc_l2_$tag.log 2>&1
    f=$(find gpurun_out/pmc_l2/$tag -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$tag" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'gemm' in r['Kernel_Name']]
agg = collections.defaultdict(float); n = collections.Counter()
for r in rows: agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
h, m = agg['TCC_HIT_sum'] / max(n['TCC_HIT_sum'], 1), agg['TCC_MISS_sum'] / max(n['TCC_MISS_sum'], 1)
print(f"{sys.argv[2]:32s} {sorted({r['Kernel_Name'][28:70] for r in rows})} hits {h:12.0f} misses {m:12.0f} hit rate {h / max(h + m, 1):.3f} requests x 128 B = {(h + m) * 128 / 1e6:8.1f} MB")
PY
  done
done
