#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4t
timeout 120 python tools/gpu/att_pipe_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4t/att_pipe_probe.txt
