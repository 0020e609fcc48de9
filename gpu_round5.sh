#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_x3.py -m gpu -q -s --tb=short --timeout=400 -k "conv or end_to_end" > gpurun_out/x3.log 2>&1
echo "exit $?" >> gpurun_out/x3.log; grep -E "bf16x3|tap |passed|failed|Error|assert " gpurun_out/x3.log | head -60
