#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 300 python tools/vp_modes_time.py bf16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/vp_modes_time.txt
timeout 900 python -m pytest tests/test_gpu_bf16a.py tests/test_gpu_bf16_modes.py tests/test_gpu_vp_engine.py -q -x 2>&1 | tail -6
bash tools/gpu_prof_mode.sh bf16 r06b > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/r06b_vp_train_b4096_bf16_kernel_stats.csv")))
print("total kernel ms/step", sum(float(r["TotalDurationNs"]) for r in rows)/5e6)
for r in rows[:16]:
    print("%-90s %6d %8.3f ms/step %8.1f us" % (r["Name"][:90], int(r["Calls"]), float(r["TotalDurationNs"])/5e6, float(r["AverageNs"])/1e3))
PY
