#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 300 python tools/gemm_bench.py 2>&1 | grep -E "TF/s"
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short --timeout=400 2>&1 | tail -2
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['kernel_breakdown'])"
