#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short --timeout=300 > gpurun_out/ops2.log 2>&1
echo "ops exit $?" >> gpurun_out/ops2.log; tail -5 gpurun_out/ops2.log
timeout 300 python tools/gemm_bench.py > gpurun_out/gemm_glds.log 2>&1; tail -30 gpurun_out/gemm_glds.log
DPTX_GEMM=reg timeout 300 python tools/gemm_bench.py > gpurun_out/gemm_reg.log 2>&1; tail -30 gpurun_out/gemm_reg.log
timeout 600 python bench.py --steps 10 --warmup 3 --profile-dump gpurun_out/launches.csv > gpurun_out/bench2.log 2>&1
echo "bench exit $?" >> gpurun_out/bench2.log; tail -3 gpurun_out/bench2.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -s --tb=short --timeout=600 > gpurun_out/e2e2.log 2>&1
echo "e2e exit $?" >> gpurun_out/e2e2.log; grep -E "max\|d\||angular|aligned|tap |passed|failed|worst|Error" gpurun_out/e2e2.log | tail -80
