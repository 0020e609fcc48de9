#!/bin/bash
# end-of-round evidence: full gpu suite, smoke, bench, rocprofv3 trace + PMC
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short --timeout=600 > gpurun_out/final/pytest_gpu.log 2>&1; echo "exit $?" >> gpurun_out/final/pytest_gpu.log; tail -4 gpurun_out/final/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/final/smoke.log 2>&1; tail -2 gpurun_out/final/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 --profile-dump gpurun_out/final/launches.csv > gpurun_out/final/bench.log 2>&1; tail -1 gpurun_out/final/bench.log | cut -c1-260
timeout 300 python bench.py --steps 10 --warmup 3 --dtype bf16x3 --no-cpu-baseline > gpurun_out/final/bench_x3.log 2>&1; tail -1 gpurun_out/final/bench_x3.log | cut -c1-200
timeout 300 python bench.py --steps 10 --warmup 3 --dtype fp16 --no-cpu-baseline > gpurun_out/final/bench_fp16.log 2>&1; tail -1 gpurun_out/final/bench_fp16.log | cut -c1-200
timeout 300 python bench.py --steps 10 --warmup 3 --task depth --no-cpu-baseline > gpurun_out/final/bench_depth.log 2>&1; tail -1 gpurun_out/final/bench_depth.log | cut -c1-200
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/trace -o r -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 1 > $R/gpurun_out/final/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/final/pmc_fetch -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 > $R/gpurun_out/final/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/final/pmc_write -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 > $R/gpurun_out/final/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS -d $R/gpurun_out/final/pmc_sq -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 > $R/gpurun_out/final/pmc_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/final/pmc_misc -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 > $R/gpurun_out/final/pmc_misc.log 2>&1
cd $R; du -sh gpurun_out/final
