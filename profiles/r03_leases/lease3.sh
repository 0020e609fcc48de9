#!/bin/bash
# round 3, GPU call 3: LN-fold row table, vectorised stem patch load, fixed tests; A/B lines + stream-count check
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c
mkdir -p $O
export TMPDIR=/tmp
python -c "from omnidata_amd.engine import load_library; print(load_library().dptx_version())" > $O/version.log 2>&1; cat $O/version.log
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py tests/test_gpu_mixed.py -m gpu -q --tb=short --timeout=900 --maxfail=8 > $O/pytest_gpu.log 2>&1; echo "exit $?" >> $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log
line() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'], d['kernel_breakdown'], (d.get('parity') or {}).get('parity_mode'))"; }
for rep in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --profile-dump $O/launches_fold$rep.csv > $O/bench_fold$rep.log 2>&1; line $O/bench_fold$rep.log fold$rep
DPTX_LN_FOLD=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --profile-dump $O/launches_nofold$rep.csv > $O/bench_nofold$rep.log 2>&1; line $O/bench_nofold$rep.log nofold$rep
done
DPTX_STREAMS=3 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also > $O/bench_3streams.log 2>&1; line $O/bench_3streams.log 3streams
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --dtype mixed > $O/bench_mixed.log 2>&1; line $O/bench_mixed.log mixed
DPTX_LN_FOLD=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --dtype mixed > $O/bench_mixed_nofold.log 2>&1; line $O/bench_mixed_nofold.log mixed_nofold
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --dtype fp16 > $O/bench_fp16.log 2>&1; line $O/bench_fp16.log fp16
du -sh $O
