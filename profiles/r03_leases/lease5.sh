#!/bin/bash
# round 3, GPU call 5: 16-bit token stream in the single-pass dtypes -- validation + A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3e
mkdir -p $O
export TMPDIR=/tmp
python -c "from omnidata_amd.engine import load_library; print(load_library().dptx_version())" > $O/version.log 2>&1; cat $O/version.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short --timeout=900 --maxfail=8 > $O/pytest_gpu.log 2>&1; echo "exit $?" >> $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log
timeout 300 python -m pytest tests/test_gpu_e2e.py -m gpu -q -s -k "layernorm_fold or engine_vs_oracle" > $O/pytest_print.log 2>&1; grep -E "^\[|^    \[|max\|d\|" $O/pytest_print.log | head -40
line() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'], d['kernel_breakdown']['gemm']['ms_per_step'], (d.get('parity') or {}).get('benched_dtype'))"; }
for rep in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-also --parity-dtype none --profile-dump $O/launches_s16_$rep.csv > $O/bench_s16_$rep.log 2>&1; line $O/bench_s16_$rep.log s16_$rep
DPTX_STREAM16=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-also --parity-dtype none --profile-dump $O/launches_s32_$rep.csv > $O/bench_s32_$rep.log 2>&1; line $O/bench_s32_$rep.log s32_$rep
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --dtype fp16 > $O/bench_fp16.log 2>&1; line $O/bench_fp16.log fp16
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --dtype fp8 > $O/bench_fp8.log 2>&1; line $O/bench_fp8.log fp8
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --task dual --dtype fp8 > $O/bench_dualfp8.log 2>&1; line $O/bench_dualfp8.log dualfp8
du -sh $O
