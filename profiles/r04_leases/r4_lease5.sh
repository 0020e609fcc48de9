mkdir -p gpurun_out/r4e
L=gpurun_out/r4e/hunt2.log
: > $L
timeout 200 python tools/gpu/r4_hunt2.py part1 0 bf16 3000 >> $L 2>&1
timeout 200 python tools/gpu/r4_hunt2.py part1 1 bf16 3000 >> $L 2>&1
timeout 200 python tools/gpu/r4_hunt2.py part1 4 bf16 3000 >> $L 2>&1
timeout 200 python tools/gpu/r4_hunt2.py part1 0 fp16 3000 >> $L 2>&1
DPTX_TILE=128128 timeout 200 python tools/gpu/r4_hunt2.py part1 0 bf16 3000 >> $L 2>&1
DPTX_TILE=6464 timeout 200 python tools/gpu/r4_hunt2.py part1 0 bf16 3000 >> $L 2>&1
timeout 300 python tools/gpu/r4_hunt2.py part2 4000 >> $L 2>&1
grep -v amdgpu.ids $L | tail -40
