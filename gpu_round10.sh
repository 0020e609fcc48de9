#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_x3.py -m gpu -q --tb=short --timeout=400 > gpurun_out/ops10.log 2>&1
echo "exit $?" >> gpurun_out/ops10.log; tail -4 gpurun_out/ops10.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-dump gpurun_out/launches10.csv > gpurun_out/bench10.log 2>&1; tail -1 gpurun_out/bench10.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['kernel_breakdown'])"
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short --timeout=400 > gpurun_out/e2e10.log 2>&1; tail -3 gpurun_out/e2e10.log
