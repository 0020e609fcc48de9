#!/bin/bash
# the race checks of section 1 once more, on the FINAL library (with the two-plane ping-pong kernel): op level and forward level
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4zz; L=gpurun_out/r4zz/verify_final.log; : > $L
python -c "from omnidata_amd.engine import load_library; print(load_library().dptx_version())" >> $L 2>&1
MICRO_QUICK=1 timeout 150 python tools/gpu/r4_micro.py 200 >> $L 2>&1
HUNT_DTYPE=fp16 timeout 200 python tools/gpu/r4_hunt3.py counts 3000 >> $L 2>&1
HUNT_DTYPE=bf16 timeout 200 python tools/gpu/r4_hunt3.py counts 3000 >> $L 2>&1
HUNT_DTYPE=mixed timeout 300 python tools/gpu/r4_hunt3.py counts 2000 >> $L 2>&1
grep -v amdgpu.ids $L
