#!/bin/bash
# round 3, GPU call 18: direct vs staged epilogue in-network after making the LayerNorm-fold / GELU arithmetic explicit fmas;
# batch invariance; the e2e + ops suites
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=$GRAFT_REPO_ROOT/gpurun_out/r3r
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/gpu/direct_probe2.py > $O/probe2.txt 2>&1; grep -v amdgpu.ids $O/probe2.txt | tail -30
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
