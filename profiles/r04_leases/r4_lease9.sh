mkdir -p gpurun_out/r4i
L=gpurun_out/r4i/micro_variants.log
: > $L
R=$PWD/omnidata_amd
for lib in "$@"; do
  MICRO_QUICK=1 DPTX_LIB=$R/$lib timeout 120 python tools/gpu/r4_micro.py 100 >> $L 2>&1
done
grep -v amdgpu.ids $L | grep "victim launches"
