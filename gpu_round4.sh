#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_x3.py -m gpu -q -s --tb=short --timeout=400 > gpurun_out/ops4.log 2>&1
echo "exit $?" >> gpurun_out/ops4.log; grep -E "bf16x3|passed|failed|Error|assert" gpurun_out/ops4.log | tail -30
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -s --tb=short --timeout=400 > gpurun_out/e2e4.log 2>&1
echo "exit $?" >> gpurun_out/e2e4.log; grep -E "passed|failed|worst|Error" gpurun_out/e2e4.log | tail -10
timeout 300 python bench.py --steps 5 --warmup 2 --dtype bf16x3 --no-cpu-baseline > gpurun_out/bench_x3.log 2>&1; tail -1 gpurun_out/bench_x3.log | cut -c1-300
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench4.log 2>&1; tail -1 gpurun_out/bench4.log | cut -c1-300
