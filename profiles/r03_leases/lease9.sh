#!/bin/bash
# round 3, GPU call 9: persistent tile loop of gemm_pp_kernel (DPTX_PERSIST = blocks per XCD; 0 = one block per tile)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3i
mkdir -p $O
export TMPDIR=/tmp
# correctness first: op-level GEMM / conv tests and the end-to-end oracle comparisons run on the persistent kernel (default)
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
SH="vit.qkv,vit.proj,vit.fc1,vit.fc2,patch.proj,cal.8192,rcu@96,rcu@48,head.0,l2_rn,l3_rn,s2.c2,pp4.conv2"
for P in 0 32; do
  DPTX_PERSIST=$P timeout 300 python tools/gemm_bench.py --only $SH > $O/shapes_p$P.txt 2>&1
done
paste <(awk '{print $1, $(NF-1)}' $O/shapes_p0.txt) <(awk '{print $(NF-1)}' $O/shapes_p32.txt)
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --parity-dtype none"
for rep in 1 2; do
  for P in 0 32 16; do
    DPTX_PERSIST=$P timeout 300 $B > $O/bf16_p${P}_$rep.log 2>&1; echo "bf16 persist=$P: $(tail -1 $O/bf16_p${P}_$rep.log | cut -c1-90)"
  done
done
for P in 0 32; do
  DPTX_PERSIST=$P timeout 300 $B --dtype mixed > $O/mixed_p$P.log 2>&1; echo "mixed persist=$P: $(tail -1 $O/mixed_p$P.log | cut -c1-90)"
  DPTX_PERSIST=$P DPTX_STREAMS=1 timeout 300 $B > $O/bf16_1s_p$P.log 2>&1; echo "bf16 1-stream persist=$P: $(tail -1 $O/bf16_1s_p$P.log | cut -c1-90)"
done
