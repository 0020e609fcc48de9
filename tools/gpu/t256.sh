#!/bin/bash
# A/B of the 256x256 8-wave GEMM tile against the 128x128 tile + correctness
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short --timeout=200 -k "gemm or conv" 2>&1 | tail -8
echo "=== 128 tile"; DPTX_TILE=128 timeout 200 python tools/gemm_bench.py --iters 20 2>&1 | tee gpurun_out/gemm_t128.log | tail -32
echo "=== 256 tile (K>=64)"; DPTX_T256_MINK=64 timeout 200 python tools/gemm_bench.py --iters 20 2>&1 | tee gpurun_out/gemm_t256.log | tail -32
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['kernel_breakdown'])"
