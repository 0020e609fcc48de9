#!/bin/bash
# round 3, GPU call 4: HEAD validation (full suite), numbers printed by the fp8 / B=32 tests, one bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3d
mkdir -p $O
export TMPDIR=/tmp
python -c "from omnidata_amd.engine import load_library; print(load_library().dptx_version())" > $O/version.log 2>&1; cat $O/version.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short --timeout=900 > $O/pytest_gpu.log 2>&1; echo "exit $?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 600 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_mixed.py -m gpu -q -s -k "families or follow_the_data or b32 or group_policy or stated_tolerance" > $O/pytest_print.log 2>&1; grep -E "^\[|^    |rms|max\|d\|" $O/pytest_print.log | head -40
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -4 $O/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-200
du -sh $O
