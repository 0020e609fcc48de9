#!/bin/bash
# round 3, GPU call 15: transposed-accumulator ping-pong kernel with the register-direct epilogue (DPTX_DIRECT=0: staged epilogue)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3o
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
DPTX_LIB=$R/omnidata_amd/libdptx_trace.so timeout 300 python tools/gpu/tile_trace.py > $O/tile_trace.txt 2>&1; grep -v amdgpu.ids $O/tile_trace.txt | cut -c1-110
SH="vit.qkv,vit.fc1,rcu@96,rcu@48,head.0,l2_rn,l3_rn"
for D in 0 1 0 1; do
  DPTX_DIRECT=$D timeout 300 python tools/gemm_bench.py --only $SH > $O/shapes_d$D.txt 2>&1; echo "direct=$D: $(grep 'TF/s' $O/shapes_d$D.txt | awk '{print $1, $(NF-1)}' | tr '\n' ';')"
done
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also"
for D in 0 1; do   # parity numbers of both forms (they must agree digit for digit: same arithmetic per element)
  DPTX_DIRECT=$D timeout 300 $B > $O/bf16_par_d$D.log 2>&1; echo "direct=$D bf16: $(tail -1 $O/bf16_par_d$D.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["parity"]["benched_dtype"]["max_abs"], d["parity"]["parity_mode"]["value"], d["parity"]["parity_mode"]["max_abs"])')"
done
for rep in 1 2 3; do
  for D in 0 1; do
    DPTX_DIRECT=$D timeout 300 $B --parity-dtype none > $O/bf16_d${D}_$rep.log 2>&1; echo "direct=$D bf16: $(tail -1 $O/bf16_d${D}_$rep.log | cut -c76-90)"
  done
done
for D in 0 1; do
  DPTX_DIRECT=$D DPTX_STREAMS=1 timeout 300 $B --parity-dtype none --profile-dump $O/launches_d$D.csv > $O/1s_d$D.log 2>&1; echo "direct=$D 1-stream: $(tail -1 $O/1s_d$D.log | cut -c76-90)"
done
