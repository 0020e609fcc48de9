#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short --timeout=400 > gpurun_out/ops6.log 2>&1
echo "exit $?" >> gpurun_out/ops6.log; tail -6 gpurun_out/ops6.log
timeout 300 python tools/gemm_bench.py > gpurun_out/gemm_3s.log 2>&1; cat gpurun_out/gemm_3s.log | grep -v amdgpu.ids
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench6.log 2>&1; tail -1 gpurun_out/bench6.log | cut -c1-200; tail -1 gpurun_out/bench6.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['achieved'], d['kernel_breakdown'])"
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short --timeout=400 -k "deterministic or oracle" > gpurun_out/e2e6.log 2>&1; tail -3 gpurun_out/e2e6.log
