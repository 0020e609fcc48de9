#!/bin/bash
# round 3, GPU call 17: locate the first layer whose result differs between the direct and the staged epilogue in-network
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=$GRAFT_REPO_ROOT/gpurun_out/r3q
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/gpu/direct_probe2.py > $O/probe2.txt 2>&1; grep -v amdgpu.ids $O/probe2.txt | tail -40
