#!/bin/bash
# round 2, lease 8: ping-pong schedule of the 256x256 tile (gemm_pp_kernel) vs the lockstep loop
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2h
mkdir -p $O
DPTX_PP=1 timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "gemm or conv" 2>&1 | tail -2
SH=vit.qkv,vit.proj,vit.fc1,vit.fc2,rcu@96,rcu@48,head.0,l2_rn,l3_rn,patch.proj
DPTX_PP=0 timeout 200 python tools/gemm_bench.py --iters 20 --only $SH > $O/gemm_pp0.log 2>&1
DPTX_PP=1 timeout 200 python tools/gemm_bench.py --iters 20 --only $SH > $O/gemm_pp1.log 2>&1
DPTX_PP=0 DPTX_T256_MINK=512 timeout 200 python tools/gemm_bench.py --iters 20 --only $SH > $O/gemm_pp0_k512.log 2>&1
DPTX_PP=1 DPTX_T256_MINK=512 timeout 200 python tools/gemm_bench.py --iters 20 --only $SH > $O/gemm_pp1_k512.log 2>&1
echo "shape lockstep pingpong lockstep_k512 pingpong_k512"
paste <(grep TF $O/gemm_pp0.log | awk '{print $1, $(NF-1)}') <(grep TF $O/gemm_pp1.log | awk '{print $(NF-1)}') <(grep TF $O/gemm_pp0_k512.log | awk '{print $(NF-1)}') <(grep TF $O/gemm_pp1_k512.log | awk '{print $(NF-1)}')
DPTX_PP=1 timeout 100 python tools/gpu/gemm_trace.py 2>&1 | grep -v amdgpu.ids > $O/trace_pp1.log; grep -A3 "rcu@96 bf16" $O/trace_pp1.log
for pp in 0 1; do
DPTX_PP=$pp timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_pp$pp.log 2>&1; tail -1 $O/bench_pp$pp.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pp$pp', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['kernel_breakdown']['gemm'])"
done
