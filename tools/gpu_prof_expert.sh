#!/bin/bash
# rocprofv3 kernel stats of the MPC-expert search (tools/expert_bench.py workload).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/prof_expert; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_expert -- python3 tools/expert_bench.py > gpurun_out/prof_expert.log 2>&1; echo "rc=$?"
f=$(ls -t gpurun_out/prof_expert/*/*kernel_stats.csv | head -1); head -8 "$f" | cut -c1-160
