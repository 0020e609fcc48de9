#!/bin/bash
# round 3, GPU call 13: wave-private vs block-wide epilogue staging, same box (libdptx_nowp.so = -DDPTX_NO_WP)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3m
mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --parity-dtype none"
SH="vit.qkv,vit.proj,vit.fc1,vit.fc2,rcu@96,head.0"
for V in nowp cur nowp cur; do
  L=""; [ $V = nowp ] && L=$R/omnidata_amd/libdptx_nowp.so
  DPTX_LIB=$L timeout 300 python tools/gemm_bench.py --only $SH > $O/shapes_$V.txt 2>&1; echo "$V: $(grep 'TF/s' $O/shapes_$V.txt | awk '{print $1, $(NF-1)}' | tr '\n' ';')"
done
for rep in 1 2; do
  for V in nowp cur; do
    L=""; [ $V = nowp ] && L=$R/omnidata_amd/libdptx_nowp.so
    DPTX_LIB=$L DPTX_STREAMS=1 timeout 300 $B --profile-dump $O/launches_${V}_$rep.csv > $O/${V}_1s_$rep.log 2>&1; echo "$V 1-stream: $(tail -1 $O/${V}_1s_$rep.log | cut -c76-90)"
    DPTX_LIB=$L timeout 300 $B > $O/${V}_$rep.log 2>&1; echo "$V bf16: $(tail -1 $O/${V}_$rep.log | cut -c76-90)"
  done
done
