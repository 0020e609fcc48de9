#!/bin/bash
# round 3, GPU call 11: wave-private epilogue staging + zero-C first k-step + balanced prologue in the persistent ping-pong kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3k
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
DPTX_LIB=$R/omnidata_amd/libdptx_trace.so timeout 300 python tools/gpu/tile_trace.py > $O/tile_trace.txt 2>&1; cat $O/tile_trace.txt | grep -v amdgpu.ids
SH="vit.qkv,vit.proj,vit.fc1,vit.fc2,patch.proj,cal.8192,rcu@96,rcu@48,head.0,l2_rn,l3_rn"
timeout 300 python tools/gemm_bench.py --only $SH > $O/shapes.txt 2>&1; grep "TF/s" $O/shapes.txt | awk '{print $1, $(NF-1)}' | tr '\n' ';'; echo
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --parity-dtype none"
for rep in 1 2; do
  timeout 300 $B > $O/bf16_$rep.log 2>&1; echo "bf16: $(tail -1 $O/bf16_$rep.log | cut -c1-90)"
done
timeout 300 $B --dtype mixed > $O/mixed.log 2>&1; echo "mixed: $(tail -1 $O/mixed.log | cut -c1-90)"
DPTX_STREAMS=1 timeout 300 $B > $O/bf16_1s.log 2>&1; echo "bf16 1-stream: $(tail -1 $O/bf16_1s.log | cut -c1-90)"
