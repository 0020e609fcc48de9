#!/bin/bash
# round 2, lease 7: DMA issue spread over the MFMAs (compile-time variants s1 / s2) vs the burst in front of them
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2g
mkdir -p $O
SH=vit.qkv,vit.proj,vit.fc1,vit.fc2,rcu@96,rcu@48,head.0,l2_rn,l3_rn,s2.c2,out_conv@96,s2.c3,pp4.conv2
for v in "" _s1 _s2; do
  L=$R/omnidata_amd/libdptx$v.so
  DPTX_LIB=$L timeout 200 python tools/gemm_bench.py --iters 20 --only $SH > $O/gemm$v.log 2>&1
  DPTX_LIB=$L timeout 100 python tools/gpu/gemm_trace.py 2>&1 | grep -v amdgpu.ids > $O/trace$v.log
done
echo "shape burst split1 split2"
paste <(grep TF $O/gemm.log | awk '{print $1, $(NF-1)}') <(grep TF $O/gemm_s1.log | awk '{print $(NF-1)}') <(grep TF $O/gemm_s2.log | awk '{print $(NF-1)}')
for v in "" _s1 _s2; do echo "--- trace$v"; grep -E "==|wave" $O/trace$v.log | head -8; done
for v in _s1 _s2; do
DPTX_LIB=$R/omnidata_amd/libdptx$v.so timeout 200 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "gemm or conv" 2>&1 | tail -1
done
