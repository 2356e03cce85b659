#!/bin/bash
# round 4: HBM-side traffic PER SHAPE of the fp32 GEMM launches of the B = 4096 train step -- FETCH_SIZE and WRITE_SIZE in
# separate PMC passes (no tracing) of each shape run alone (tools/gemm_pmc.py: 6 launches), FETCH_SIZE x2 per the gfx950 correction.
# Output: a table (algorithmic bytes = A + B + C per launch; counted bytes; ratio; launches per step) + the step-weighted totals.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out/pmc_shape
PREC=${PREC:-f32}
python3 - <<'PY' > gpurun_out/pmc_shape/shapes.txt
import sys, os
sys.path.insert(0, os.getcwd())
SH = [  # name, a_kmajor, b_kmajor, M, N, K, accumulate, launches per step (tools/gemm_bench.py)
    ('enc_qkv_fwd', 0, 0, 40960, 1536, 512, 0, 2), ('enc_512_fwd', 0, 0, 40960, 512, 512, 0, 6), ('conv_fwd', 0, 0, 40960, 512, 1536, 0, 1),
    ('memkv_fwd', 0, 0, 20480, 1024, 512, 0, 2), ('dec_qkv_fwd', 0, 0, 4096, 1536, 512, 0, 20), ('dec_512_fwd', 0, 0, 4096, 512, 512, 0, 100),
    ('dec_512_dX', 0, 1, 4096, 512, 512, 0, 100), ('dec_qkv_dX', 0, 1, 4096, 512, 1536, 0, 20), ('enc_512_dX', 0, 1, 40960, 512, 512, 0, 6),
    ('enc_qkv_dX', 0, 1, 40960, 512, 1536, 0, 2), ('conv_dX', 0, 1, 40960, 1536, 512, 0, 1),
    ('dW_512x512', 1, 1, 512, 512, 40960, 1, 16), ('dW_1536x512', 1, 1, 1536, 512, 40960, 1, 4), ('dW_conv_512x1536', 1, 1, 512, 1536, 40960, 1, 1),
    ('dW_kv_1024x512', 1, 1, 1024, 512, 20480, 1, 2)]
for s in SH: print(*s)
PY
while read name ak bk M N K acc cnt; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_shape/${name}_$c -- python3 tools/gemm_pmc.py $M $N $K $ak $bk $acc $PREC > gpurun_out/pmc_shape/${name}_$c.log 2>&1 || echo "rc=$? $name $c"
  done
done < gpurun_out/pmc_shape/shapes.txt
python3 - <<'PY'
import csv, glob, os
rows = [l.split() for l in open('gpurun_out/pmc_shape/shapes.txt')]
def per_launch(name, counter):
    f = sorted(glob.glob(f'gpurun_out/pmc_shape/{name}_{counter}/**/*counter_collection.csv', recursive=True))[-1]
    v = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if r['Counter_Name'] == counter and 'gemm_' in r['Kernel_Name']]
    k = sorted({r['Kernel_Name'].split('(')[0][-60:] for r in csv.DictReader(open(f)) if 'gemm_' in r['Kernel_Name']})
    return sum(v) / len(v) * 1024.0, k
print(f"{'shape':18s} {'M':>6s} {'N':>5s} {'K':>6s} {'/step':>5s} {'alg MB':>8s} {'fetch MB':>9s} {'write MB':>9s} {'counted/alg':>11s}  kernel")
tot_alg = tot_cnt = 0.0
for name, ak, bk, M, N, K, acc, cnt in rows:
    M, N, K, cnt, acc = int(M), int(N), int(K), int(cnt), int(acc)
    alg = 4.0 * (M * K + N * K + M * N)                  # A + B read once, C written once (accumulating dW: the atomics ARE the C write)
    fe, k = per_launch(name, 'FETCH_SIZE'); wr, _ = per_launch(name, 'WRITE_SIZE')
    fe *= 2.0                                            # gfx950: FETCH_SIZE reports half of a wide streaming read
    tot_alg += alg * cnt; tot_cnt += (fe + wr) * cnt
    print(f"{name:18s} {M:6d} {N:5d} {K:6d} {cnt:5d} {alg/1e6:8.1f} {fe/1e6:9.1f} {wr/1e6:9.1f} {(fe+wr)/alg:11.2f}  {k[0][-48:]}")
print(f"step-weighted: algorithmic {tot_alg/1e9:.2f} GB, counted {tot_cnt/1e9:.2f} GB per step, ratio {tot_cnt/tot_alg:.2f}")
PY
