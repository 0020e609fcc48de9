#!/bin/bash
# same-box A/B: default build vs build without the fp8 epilogue hooks vs ping-pong 256 tile
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2i
mkdir -p $O
for rep in 1 2; do
for v in "" _nof8; do
DPTX_LIB=$R/omnidata_amd/libdptx$v.so timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench$v.log 2>&1; tail -1 $O/bench$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$v', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['kernel_breakdown']['gemm']['ms_per_step'])"
done
DPTX_PP=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_pp.log 2>&1; tail -1 $O/bench_pp.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pp', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['kernel_breakdown']['gemm']['ms_per_step'])"
done
( for i in $(seq 1 20); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' '; echo; sleep 0.5; done ) > $O/smi.log 2>&1 &
timeout 600 python bench.py --steps 400 --warmup 5 --no-cpu-baseline > $O/bench_long.log 2>&1
wait
tail -12 $O/smi.log
