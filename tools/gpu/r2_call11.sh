#!/bin/bash
# same-box A/B: ping-pong with grouped reads vs software-pipelined reads; new 256-tile rule
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2j
mkdir -p $O
SH=vit.qkv,vit.proj,vit.fc1,vit.fc2,rcu@96,rcu@48,head.0,l2_rn,l3_rn,patch.proj
DPTX_LIB=$R/omnidata_amd/libdptx_pipe.so timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "gemm or conv" 2>&1 | tail -1
for v in "" _pipe; do
  DPTX_LIB=$R/omnidata_amd/libdptx$v.so timeout 200 python tools/gemm_bench.py --iters 30 --only $SH > $O/gemm$v.log 2>&1
done
echo "shape pp pp+pipe"
paste <(grep TF $O/gemm.log | awk '{print $1, $(NF-1)}') <(grep TF $O/gemm_pipe.log | awk '{print $(NF-1)}')
for rep in 1 2; do
for v in "" _pipe; do
DPTX_LIB=$R/omnidata_amd/libdptx$v.so timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench$v.log 2>&1; tail -1 $O/bench$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$v', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['kernel_breakdown']['gemm']['ms_per_step'])"
done
done
