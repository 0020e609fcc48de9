mkdir -p gpurun_out/r4f
L=gpurun_out/r4f/hunt3.log
: > $L
R=$PWD/omnidata_amd
timeout 300 python tools/gpu/r4_hunt3.py sums 3000 >> $L 2>&1
timeout 200 python tools/gpu/r4_hunt3.py counts 3000 >> $L 2>&1
DPTX_ATT_NT=1 timeout 200 python tools/gpu/r4_hunt3.py counts 3000 >> $L 2>&1
DPTX_LIB=$R/libdptx_epint.so timeout 200 python tools/gpu/r4_hunt3.py counts 3000 >> $L 2>&1
DPTX_LIB=$R/libdptx_dmasc1.so timeout 200 python tools/gpu/r4_hunt3.py counts 3000 >> $L 2>&1
DPTX_ATT_NT=1 DPTX_LIB=$R/libdptx_all.so timeout 200 python tools/gpu/r4_hunt3.py counts 3000 >> $L 2>&1
grep -v amdgpu.ids $L | tail -40
