#!/bin/bash
cd "$GRAFT_REPO_ROOT"
SH="vit.qkv,vit.proj,vit.fc1,vit.fc2,rcu@96,rcu@48,head.0,l2_rn,out_conv@96,patch.proj"
echo "== default"; timeout 200 python tools/gemm_bench.py --only $SH 2>&1 | grep -E "TF/s"
echo "== g32"; DPTX_GEMM=g32 timeout 200 python tools/gemm_bench.py --only $SH 2>&1 | grep -E "TF/s"
DPTX_GEMM=g32 timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short --timeout=400 -k "gemm or conv" 2>&1 | tail -2
DPTX_GEMM=g32 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['kernel_breakdown'])"
