#!/bin/bash
# round 3, GPU call 14: packed GELU + LayerNorm statistics DMA'd to LDS under the k-loop, same-box A/B against the previous build
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3n
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --parity-dtype none"
for V in prev cur prev cur; do
  L=""; [ $V = prev ] && L=$R/omnidata_amd/libdptx_prev.so
  DPTX_LIB=$L timeout 300 python tools/gemm_bench.py --only vit.qkv,vit.fc1 > $O/shapes_$V.txt 2>&1; echo "$V: $(grep 'TF/s' $O/shapes_$V.txt | awk '{print $1, $(NF-1)}' | tr '\n' ';')"
done
for rep in 1 2 3; do
  for V in prev cur; do
    L=""; [ $V = prev ] && L=$R/omnidata_amd/libdptx_prev.so
    DPTX_LIB=$L timeout 300 $B > $O/${V}_$rep.log 2>&1; echo "$V bf16: $(tail -1 $O/${V}_$rep.log | cut -c76-90)"
  done
done
for V in prev cur; do
  L=""; [ $V = prev ] && L=$R/omnidata_amd/libdptx_prev.so
  DPTX_LIB=$L DPTX_STREAMS=1 timeout 300 $B --profile-dump $O/launches_$V.csv > $O/${V}_1s.log 2>&1; echo "$V 1-stream: $(tail -1 $O/${V}_1s.log | cut -c76-90)"
  DPTX_LIB=$L timeout 300 $B --dtype mixed > $O/${V}_mixed.log 2>&1; echo "$V mixed: $(tail -1 $O/${V}_mixed.log | cut -c76-90)"
done
