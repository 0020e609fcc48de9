#!/bin/bash
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/headprof
python tools/head_bench.py
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $R/gpurun_out/headprof/a -o r -- python $R/tools/head_bench.py --iters 5 > $R/gpurun_out/headprof/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d $R/gpurun_out/headprof/b -o r -- python $R/tools/head_bench.py --iters 5 > $R/gpurun_out/headprof/b.log 2>&1
cd $R
for d in a b; do f=$(find gpurun_out/headprof/$d -name "*.db" | head -1); python tools/rocprof_summary.py $f --pmc 2>&1 | grep -i "head_tail" ; done
tail -3 gpurun_out/headprof/b.log
