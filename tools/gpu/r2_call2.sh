#!/bin/bash
# round 2, lease 2: new GEMM epilogue (batched residual loads, GroupNorm statistics), reciprocal row division
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2b
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_x3.py tests/test_gpu_mixed.py -m gpu -q --tb=short --timeout=600 -x > $O/pytest_ops.log 2>&1; echo "exit $?" >> $O/pytest_ops.log; grep -E "passed|failed|exit|FAILED|Error" $O/pytest_ops.log | tail -8
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_dual.py tests/test_gpu_flex.py -m gpu -q --tb=short --timeout=600 > $O/pytest_e2e.log 2>&1; echo "exit $?" >> $O/pytest_e2e.log; grep -E "passed|failed|exit|FAILED|Error" $O/pytest_e2e.log | tail -8
timeout 300 python tools/gemm_bench.py --iters 20 > $O/gemm_bench.log 2>&1; cat $O/gemm_bench.log | tail -30
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-dump $O/launches.csv > $O/bench.log 2>&1; tail -1 $O/bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['kernel_breakdown'])"
DPTX_GN_FUSED=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_nofuse.log 2>&1; tail -1 $O/bench_nofuse.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nofuse', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['kernel_breakdown'])"
DPTX_STREAMS=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_1s.log 2>&1; tail -1 $O/bench_1s.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('1stream', d['value'], d['ms_per_step'])"
