#!/bin/bash
# round 2, lease 4: attention x3 diagnosis, DMA cache policy, 8-wave x3 tile, 256 tile for short K
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2d
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python tools/gpu/att_debug.py > $O/att_debug.log 2>&1; cat $O/att_debug.log | tail -20
SH=vit.qkv,vit.proj,vit.fc1,vit.fc2,rcu@96,rcu@48,head.0,l2_rn,l3_rn,s2.c2,out_conv@96,s2.c3,pp4.conv2
for a in 0 1 2 3; do
  DPTX_DMA_AUX=$a timeout 200 python tools/gemm_bench.py --iters 20 --only $SH > $O/gemm_aux$a.log 2>&1
done
DPTX_T256_MINK=512 timeout 200 python tools/gemm_bench.py --iters 20 --only $SH > $O/gemm_t256k512.log 2>&1
echo "shape aux0 auxA auxW auxAW t256k512"
paste <(grep TF $O/gemm_aux0.log | awk '{print $1, $(NF-1)}') <(grep TF $O/gemm_aux1.log | awk '{print $(NF-1)}') <(grep TF $O/gemm_aux2.log | awk '{print $(NF-1)}') <(grep TF $O/gemm_aux3.log | awk '{print $(NF-1)}') <(grep TF $O/gemm_t256k512.log | awk '{print $(NF-1)}')
for w in 0 1; do
DPTX_X3_W8=$w timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype fp16x3 > $O/bench_x3w$w.log 2>&1; tail -1 $O/bench_x3w$w.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('x3 w8=$w', d['value'], d['ms_per_step'], d['kernel_breakdown'])"
done
DPTX_X3_W8=1 timeout 300 python -m pytest tests/test_gpu_x3.py tests/test_gpu_mixed.py -m gpu -q -x -k "gemm or conv" > $O/pytest_w8.log 2>&1; tail -2 $O/pytest_w8.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.log 2>&1; tail -1 $O/bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('base', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['kernel_breakdown'])"
