#!/bin/bash
# round 3, GPU call 12: same-box A/B of the persistent / wave-private-epilogue ping-pong kernel against the library of commit
# f80fc9d (libdptx_base.so, built from that commit's csrc)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3l
mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --parity-dtype none"
for rep in 1 2 3; do
  DPTX_LIB=$R/omnidata_amd/libdptx_base.so timeout 300 $B > $O/base_$rep.log 2>&1; echo "base   bf16: $(tail -1 $O/base_$rep.log | cut -c76-90)"
  timeout 300 $B > $O/cur_$rep.log 2>&1; echo "current bf16: $(tail -1 $O/cur_$rep.log | cut -c76-90)"
done
for D in mixed fp16; do
  DPTX_LIB=$R/omnidata_amd/libdptx_base.so timeout 300 $B --dtype $D > $O/base_$D.log 2>&1; echo "base   $D: $(tail -1 $O/base_$D.log | cut -c76-90)"
  timeout 300 $B --dtype $D > $O/cur_$D.log 2>&1; echo "current $D: $(tail -1 $O/cur_$D.log | cut -c76-90)"
done
DPTX_LIB=$R/omnidata_amd/libdptx_base.so DPTX_STREAMS=1 timeout 300 $B > $O/base_1s.log 2>&1; echo "base   1-stream: $(tail -1 $O/base_1s.log | cut -c76-90)"
DPTX_STREAMS=1 timeout 300 $B --profile-dump $O/launches.csv > $O/cur_1s.log 2>&1; echo "current 1-stream: $(tail -1 $O/cur_1s.log | cut -c76-90)"
DPTX_LIB=$R/omnidata_amd/libdptx_base.so timeout 300 $B --backbone vitl16_384 --task depth > $O/base_L.log 2>&1; echo "base   DPT-Large: $(tail -1 $O/base_L.log | cut -c70-95)"
timeout 300 $B --backbone vitl16_384 --task depth > $O/cur_L.log 2>&1; echo "current DPT-Large: $(tail -1 $O/cur_L.log | cut -c70-95)"
