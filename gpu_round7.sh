#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
SH="vit.fc1,rcu@96,head.0,head.2,vit.qkv"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS -d $R/gpurun_out/pmc/sq -o r -- python $R/tools/gemm_bench.py --only $SH --iters 3 > $R/gpurun_out/pmc/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM -d $R/gpurun_out/pmc/lds -o r -- python $R/tools/gemm_bench.py --only $SH --iters 3 > $R/gpurun_out/pmc/lds.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d $R/gpurun_out/pmc/tcc -o r -- python $R/tools/gemm_bench.py --only $SH --iters 3 > $R/gpurun_out/pmc/tcc.log 2>&1
cd $R; tail -3 gpurun_out/pmc/sq.log; tail -3 gpurun_out/pmc/lds.log; tail -3 gpurun_out/pmc/tcc.log; ls -la gpurun_out/pmc/*/
