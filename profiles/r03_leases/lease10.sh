#!/bin/bash
# round 3, GPU call 10: where a tile's time goes in the persistent ping-pong kernel (cycle stamps, trace build) + the 7.1.28 GELU
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3j
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm or gelu or readout" > $O/tests.log 2>&1; tail -2 $O/tests.log
DPTX_LIB=$R/omnidata_amd/libdptx_trace.so timeout 300 python tools/gpu/tile_trace.py > $O/tile_trace.txt 2>&1; cat $O/tile_trace.txt | grep -v amdgpu.ids
timeout 300 python tools/gemm_bench.py --only vit.qkv,vit.proj,vit.fc1,vit.fc2 > $O/shapes.txt 2>&1; grep vit $O/shapes.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --parity-dtype none"
for rep in 1 2; do
  timeout 300 $B > $O/bf16_$rep.log 2>&1; echo "bf16: $(tail -1 $O/bf16_$rep.log | cut -c1-90)"
done
timeout 300 $B --dtype mixed > $O/mixed.log 2>&1; echo "mixed: $(tail -1 $O/mixed.log | cut -c1-90)"
