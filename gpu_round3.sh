#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short --timeout=300 -k "gemm_epilogues" > gpurun_out/ops3.log 2>&1; tail -3 gpurun_out/ops3.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench3.log 2>&1; tail -2 gpurun_out/bench3.log | cut -c1-400
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/trace -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 1 > $GRAFT_REPO_ROOT/gpurun_out/prof/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_fetch -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 > $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_write -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 > $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof -type f | head -30; du -sh gpurun_out/prof
