mkdir -p gpurun_out/r4g
L=gpurun_out/r4g/hunt3.log
: > $L
R=$PWD/omnidata_amd
HUNT_FLAGS=1 timeout 200 python tools/gpu/r4_hunt3.py counts 3000 >> $L 2>&1
DPTX_LIB=$R/libdptx_stsc1.so timeout 200 python tools/gpu/r4_hunt3.py counts 3000 >> $L 2>&1
DPTX_LIB=$R/libdptx_ldsys.so timeout 200 python tools/gpu/r4_hunt3.py counts 3000 >> $L 2>&1
DPTX_LIB=$R/libdptx_both.so timeout 200 python tools/gpu/r4_hunt3.py counts 3000 >> $L 2>&1
timeout 200 python tools/gpu/r4_hunt3.py counts 3000 >> $L 2>&1
timeout 400 python tools/gpu/r4_hunt3.py sums 4000 >> $L 2>&1
grep -v amdgpu.ids $L | tail -40
