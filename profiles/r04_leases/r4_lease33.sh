#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4t
timeout 200 python tools/gpu/att_probe.py 2>&1 | grep -v amdgpu.ids | grep -E "library|mask  0|mask 256" | tee gpurun_out/r4t/att_probe2.txt
