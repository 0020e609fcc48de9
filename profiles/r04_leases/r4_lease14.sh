mkdir -p gpurun_out/r4n
L=gpurun_out/r4n/verify.log
: > $L
R=$PWD/omnidata_amd
MICRO_QUICK=1 DPTX_LIB=$R/libdptx_packed.so timeout 120 python tools/gpu/r4_micro.py 100 >> $L 2>&1
MICRO_QUICK=1 timeout 120 python tools/gpu/r4_micro.py 200 >> $L 2>&1
HUNT_DTYPE=fp16 timeout 200 python tools/gpu/r4_hunt3.py counts 3000 >> $L 2>&1
HUNT_DTYPE=bf16 timeout 200 python tools/gpu/r4_hunt3.py counts 3000 >> $L 2>&1
grep -v amdgpu.ids $L | grep "victim launches\|counts"
timeout 1500 python -m pytest tests -m gpu -q --tb=short --timeout=900 > gpurun_out/r4n/pytest_full.log 2>&1
tail -15 gpurun_out/r4n/pytest_full.log
