#!/bin/bash
# round 3, GPU call 16: is the MFMA bit-symmetric under operand swap?  (the register-direct epilogue rests on it)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=$GRAFT_REPO_ROOT/gpurun_out/r3p
mkdir -p $O
export TMPDIR=/tmp
timeout 120 tools/gpu/probes/mfma_swap > $O/mfma_swap.txt 2>&1; cat $O/mfma_swap.txt
timeout 300 python tools/gpu/direct_probe.py > $O/direct_probe.txt 2>&1; grep -v amdgpu.ids $O/direct_probe.txt
