#!/bin/bash
# end-of-round evidence (round 4): ONE full gpu suite on the final library (no -x), smoke, bench lines, rocprofv3 trace + PMC passes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${FINAL_DIR:-final4}
mkdir -p $O
export TMPDIR=/tmp
python -c "from omnidata_amd.engine import load_library; print(load_library().dptx_version())" > $O/version.log 2>&1; cat $O/version.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short --timeout=900 > $O/pytest_gpu.log 2>&1; echo "exit $?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -4 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 --profile-dump $O/launches.csv > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-300
for cfg in "--dtype mixed" "--dtype fp16x3" "--dtype fp16" "--task depth" "--task dual" "--task dual --dtype fp8" "--task dual --dtype mixed" "--dtype fp8"; do
  n=$(echo $cfg | tr -d ' -' )
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also $cfg > $O/bench_$n.log 2>&1; tail -1 $O/bench_$n.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'])"
done
timeout 300 python bench.py --dtype mixed --steps 10 --warmup 3 --no-cpu-baseline --no-also --parity-dtype none --profile-dump $O/launches_mixed.csv > /dev/null 2>&1
DPTX_STREAMS=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also > $O/bench_1stream.log 2>&1; tail -1 $O/bench_1stream.log | cut -c1-120
# same-box A/B of the two-plane ping-pong kernel (bit-identical results: tests/test_gpu_mixed.py)
for V in 0 1 0 1; do
  DPTX_PP2=$V timeout 300 python bench.py --dtype mixed --steps 10 --warmup 3 --no-cpu-baseline --no-also --parity-dtype none > $O/ab_pp2_$V.log 2>&1; echo "DPTX_PP2=$V: $(tail -1 $O/ab_pp2_$V.log | cut -c76-90)" | tee -a $O/ab_pp2.txt
done
timeout 400 python bench.py --backbone vitl16_384 --task depth --steps 8 --warmup 3 --no-also > $O/bench_vitl16.log 2>&1; tail -1 $O/bench_vitl16.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('vitl16', d['value'], d['roofline']['frac'], d['parity'])"
timeout 300 python tools/gemm_bench.py --only cal.4096,cal.8192,vit.qkv,vit.proj,vit.fc1,vit.fc2,rcu@96,rcu@48,head.0,l2_rn,l3_rn,s2.c1,s2.c2,s2.c3 --iters 30 > $O/gemm_shapes.txt 2>&1; grep TF/s $O/gemm_shapes.txt | tail -16
timeout 200 python tools/gpu/r4_headx3_bench.py > $O/headx3_bench.txt 2>&1; tail -3 $O/headx3_bench.txt
cd /tmp
export DPTX_STREAMS=1   # kernel-level passes: one launch per layer over the whole batch (the bench's per-launch figures)
B="python $R/bench.py --no-cpu-baseline --no-also --parity-dtype none --profile-steps 1"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- $B --steps 5 --warmup 2 > $O/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_mixed -o r -- $B --dtype mixed --steps 3 --warmup 1 > $O/trace_mixed.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o r -- $B --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o r -- $B --steps 2 --warmup 1 > $O/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS -d $O/pmc_sq -o r -- $B --steps 2 --warmup 1 > $O/pmc_sq.log 2>&1
unset DPTX_STREAMS
cd $R
db() { find $O/$1 -name "*.db" | head -1; }
python tools/rocprof_summary.py $(db trace) > $O/r04_kernel_trace_stats.txt 2>&1
python tools/rocprof_summary.py $(db trace_mixed) > $O/r04_kernel_trace_stats_mixed.txt 2>&1
python tools/rocprof_summary.py $(db pmc_fetch) --pmc > $O/r04_pmc_fetch_size.txt 2>&1
python tools/rocprof_summary.py $(db pmc_write) --pmc > $O/r04_pmc_write_size.txt 2>&1
python tools/rocprof_summary.py $(db pmc_sq) --pmc > $O/r04_pmc_sq.txt 2>&1
python tools/pmc_traffic.py $(db pmc_fetch) $(db pmc_write) 4 130 > $O/r04_pmc_traffic.json 2>&1
cat $O/r04_pmc_traffic.json | head -12
head -14 $O/r04_kernel_trace_stats.txt
find $O -name "*.db" -size +20M -delete
du -sh $O
