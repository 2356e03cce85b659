#!/bin/bash
# round 4: does a column-group tile order cut the A over-fetch of the wide-N fp32 products?  FETCH_SIZE (x2, gfx950) and the
# kernel duration of each shape alone, for group widths 0 (row-panel-major: every column tile of a panel, then the next panel) / 8 / 12.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out/pmc_cg
for shape in "40960 1536 512 0 0" "40960 1536 512 0 1" "4096 1536 512 0 0" "40960 512 1536 0 0"; do
for G in 0 8 12; do
  tag=g${G}_$(echo $shape | tr ' ' '_')
  MANSY_COL_GROUP=$G rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_cg/$tag -- python3 tools/gemm_pmc.py $shape > gpurun_out/pmc_cg/$tag.log 2>&1 || echo "rc=$? $tag"
  python3 - "$tag" "$shape" "$G" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
f = sorted(glob.glob(f'gpurun_out/pmc_cg/{tag}/**/*counter_collection.csv', recursive=True))[-1]
k = sorted(glob.glob(f'gpurun_out/pmc_cg/{tag}/**/*kernel_trace.csv', recursive=True))[-1]
v = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if r['Counter_Name'] == 'FETCH_SIZE' and 'gemm_' in r['Kernel_Name']]
d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(k)) if 'gemm_' in r['Kernel_Name']]
print(f"shape {sys.argv[2]:22s} col_group {sys.argv[3]:>2s}: fetch {2 * 1024 * sum(v) / len(v) / 1e6:8.1f} MB per launch, duration {sum(d) / len(d) / 1e3:7.1f} us (profiled, {len(d)} launches)")
PY
done; done
