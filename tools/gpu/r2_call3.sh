#!/bin/bash
# round 2, lease 3: L2 prefetch distance sweep on the GEMM shapes, attention x3 diagnosis, fat GroupNorm apply blocks
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2c
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python tools/gpu/att_debug.py > $O/att_debug.log 2>&1; cat $O/att_debug.log | tail -14
SH=vit.qkv,vit.proj,vit.fc1,vit.fc2,rcu@96,rcu@48,head.0,l2_rn,l3_rn,s2.c2,out_conv@96,s2.c3,pp4.conv2
for pf in 0 1 2 3 4; do
  DPTX_PF=$pf timeout 200 python tools/gemm_bench.py --iters 20 --only $SH > $O/gemm_pf$pf.log 2>&1
done
paste <(grep TF $O/gemm_pf0.log | awk '{print $1, $(NF-1)}') <(grep TF $O/gemm_pf1.log | awk '{print $(NF-1)}') <(grep TF $O/gemm_pf2.log | awk '{print $(NF-1)}') <(grep TF $O/gemm_pf3.log | awk '{print $(NF-1)}') <(grep TF $O/gemm_pf4.log | awk '{print $(NF-1)}')
timeout 200 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "gemm or conv" > $O/pytest_pf0.log 2>&1; tail -2 $O/pytest_pf0.log
DPTX_PF=3 timeout 200 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "gemm or conv" > $O/pytest_pf3.log 2>&1; tail -2 $O/pytest_pf3.log
for pf in 0 3; do
DPTX_PF=$pf timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --parity-dtype none > $O/bench_pf$pf.log 2>&1; tail -1 $O/bench_pf$pf.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pf$pf', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['kernel_breakdown'])"
done
