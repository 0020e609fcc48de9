#!/bin/bash
# end-of-round evidence: full gpu suite, smoke, bench lines, rocprofv3 trace + PMC passes, summaries
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short --timeout=600 > $O/pytest_gpu.log 2>&1; echo "exit $?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 --profile-dump $O/launches.csv > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-260
timeout 300 python bench.py --steps 10 --warmup 3 --dtype bf16x3 --no-cpu-baseline > $O/bench_x3.log 2>&1; tail -1 $O/bench_x3.log | cut -c1-200
timeout 300 python bench.py --steps 10 --warmup 3 --dtype fp16 --no-cpu-baseline > $O/bench_fp16.log 2>&1; tail -1 $O/bench_fp16.log | cut -c1-200
timeout 300 python bench.py --steps 10 --warmup 3 --task depth --no-cpu-baseline > $O/bench_depth.log 2>&1; tail -1 $O/bench_depth.log | cut -c1-200
timeout 300 python bench.py --steps 10 --warmup 3 --task dual --no-cpu-baseline > $O/bench_dual.log 2>&1; tail -1 $O/bench_dual.log | cut -c1-200
DPTX_STREAMS=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_1stream.log 2>&1; tail -1 $O/bench_1stream.log | cut -c1-120
cd /tmp
export DPTX_STREAMS=1   # kernel-level passes: one launch per layer over the whole batch (the bench's per-launch figures)
B="python $R/bench.py --no-cpu-baseline --profile-steps 1"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- $B --steps 5 --warmup 2 > $O/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o r -- $B --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o r -- $B --steps 2 --warmup 1 > $O/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS -d $O/pmc_sq -o r -- $B --steps 2 --warmup 1 > $O/pmc_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_misc -o r -- $B --steps 2 --warmup 1 > $O/pmc_misc.log 2>&1
unset DPTX_STREAMS
cd $R
db() { find $O/$1 -name "*.db" | head -1; }
python tools/rocprof_summary.py $(db trace) > $O/r01_kernel_trace_stats.txt 2>&1
python tools/rocprof_summary.py $(db pmc_fetch) --pmc > $O/r01_pmc_fetch_size.txt 2>&1
python tools/rocprof_summary.py $(db pmc_write) --pmc > $O/r01_pmc_write_size.txt 2>&1
python tools/rocprof_summary.py $(db pmc_sq) --pmc > $O/r01_pmc_sq.txt 2>&1
python tools/rocprof_summary.py $(db pmc_misc) --pmc > $O/r01_pmc_misc.txt 2>&1
python tools/pmc_traffic.py $(db pmc_fetch) $(db pmc_write) 4 130 > $O/r01_pmc_traffic.json 2>&1
cat $O/r01_pmc_traffic.json | head -12
find $O -name "*.db" -size +20M -delete
du -sh $O
