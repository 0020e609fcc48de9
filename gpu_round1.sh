#!/bin/bash
# first GPU session: op tests, e2e tests, smoke, bench, kernel trace
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -8 > gpurun_out/rocminfo.txt 2>&1
nproc >> gpurun_out/rocminfo.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short --timeout=300 > gpurun_out/ops.log 2>&1
echo "ops exit $?" >> gpurun_out/ops.log
tail -30 gpurun_out/ops.log
timeout 1200 python -m pytest tests/test_gpu_e2e.py -m gpu -q -s --tb=short --timeout=600 > gpurun_out/e2e.log 2>&1
echo "e2e exit $?" >> gpurun_out/e2e.log
tail -60 gpurun_out/e2e.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1
echo "bench exit $?" >> gpurun_out/bench.log
tail -5 gpurun_out/bench.log
