#!/bin/bash
# round 2, lease 5: fp8 decoder, 16-bit caller I/O, fixed fp16x3 attention, stress test, bench lines
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2e
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_mixed.py tests/test_gpu_stress.py -m gpu -q --tb=short --timeout=600 -s > $O/pytest_new.log 2>&1; echo "exit $?" >> $O/pytest_new.log; grep -E "passed|failed|exit|FAILED|Error|fp8|tap " $O/pytest_new.log | tail -40
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short --timeout=600 -k "sixteen or oracle" > $O/pytest_io.log 2>&1; tail -3 $O/pytest_io.log
for cfg in "--dtype bf16 --task dual" "--dtype fp8 --task dual" "--dtype fp8" "--dtype bf16 --io bf16"; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $cfg > $O/bench_tmp.log 2>&1; tail -1 $O/bench_tmp.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['kernel_breakdown']['gemm'])"
done
SH=rcu@96,rcu@48,rcu@24,head.0,out_conv@96
timeout 200 python tools/gemm_bench.py --iters 20 --only $SH > $O/gemm_bf16.log 2>&1; grep TF $O/gemm_bf16.log
timeout 200 python tools/gemm_bench.py --iters 20 --only $SH --dtype fp8 > $O/gemm_fp8.log 2>&1; grep -E "TF|rror" $O/gemm_fp8.log | tail -8
