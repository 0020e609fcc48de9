#!/bin/bash
# round 3, GPU call 2: per-layer precision policy of the mixed dtype, fp8 scales, LayerNorm-fold fix -- validation + A/B lines
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3b
mkdir -p $O
export TMPDIR=/tmp
python -c "from omnidata_amd.engine import load_library; print(load_library().dptx_version())" > $O/version.log 2>&1; cat $O/version.log
timeout 1700 python -m pytest tests -m gpu -q --tb=short --timeout=900 --maxfail=8 > $O/pytest_gpu.log 2>&1; echo "exit $?" >> $O/pytest_gpu.log; tail -25 $O/pytest_gpu.log
grep -E "^\[|^    \[|max\|d\||tap " $O/pytest_gpu.log | head -5
line() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'], d['kernel_breakdown'], (d.get('parity') or {}).get('parity_mode'))"; }
timeout 600 python bench.py --steps 20 --warmup 5 --profile-dump $O/launches.csv > $O/bench.log 2>&1; line $O/bench.log default
DPTX_LN_FOLD=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --profile-dump $O/launches_nofold.csv > $O/bench_nofold.log 2>&1; line $O/bench_nofold.log nofold
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --dtype mixed --profile-dump $O/launches_mixed.csv > $O/bench_mixed.log 2>&1; line $O/bench_mixed.log mixed
timeout 300 python bench.py --steps 10 --warmup 3 --no-also --dtype fp8 --parity-dtype none > $O/bench_fp8.log 2>&1; line $O/bench_fp8.log fp8
timeout 400 python tools/precision_frontier.py --steps 8 --out $O/frontier.md > $O/frontier.log 2>&1; tail -12 $O/frontier.md
du -sh $O
