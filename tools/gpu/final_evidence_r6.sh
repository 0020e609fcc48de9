#!/bin/bash
# end-of-round evidence (round 6): ONE full gpu suite on the final library (no -x), smoke, the bench lines, rocprofv3 trace + PMC
# passes.  Kernel-level passes (trace, PMC) run ONE forward at a time on ONE stream (--inflight 1, DPTX_STREAMS=1): uncontended
# kernel durations.  SKIP_SUITE=1 skips the test suite; FINAL_DIR names the output directory under gpurun_out/.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${FINAL_DIR:-final6}
mkdir -p $O
export TMPDIR=/tmp
python -c "from omnidata_amd.engine import load_library; print(load_library().dptx_version())" > $O/version.log 2>&1; cat $O/version.log
if [ -z "$SKIP_SUITE" ]; then
timeout 2400 python -m pytest tests -m gpu -q --tb=short --timeout=900 > $O/pytest_gpu.log 2>&1; echo "exit $?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
fi
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -5 $O/smoke.log
# the driver's own call first (default flags), with the in-run traffic measurement stamped with this library's hash
timeout 1200 python bench.py --steps 20 --warmup 5 --profile-dump $O/launches.csv --measure-traffic --traffic-out $O/r06_pmc_traffic.json > $O/bench.log 2>$O/bench.err; tail -1 $O/bench.log | cut -c1-300
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'value', d['value'], 'inflight1', d['value_inflight1'], 'inflight2', d['value_inflight2'], 'schedule', d['config']['engine_schedule'], 'sclk', (d['telemetry'].get('inflight1') or {}).get('sclk_mhz'))"; }
for cfg in "--dtype mixed" "--dtype fp16" "--task depth" "--task dual" "--task dual --dtype fp8" "--task dual --dtype mixed" "--inflight 2"; do
  n=$(echo $cfg | tr -d ' -' )
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --parity-dtype none $cfg > $O/bench_$n.log 2>&1; tail -1 $O/bench_$n.log | pick "$cfg"
done
timeout 300 python bench.py --dtype mixed --steps 5 --warmup 2 --no-cpu-baseline --no-also --parity-dtype none --no-schedule-ab --profile-dump $O/launches_mixed.csv > /dev/null 2>&1
DPTX_STREAMS=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --parity-dtype none --no-schedule-ab > $O/bench_1stream.log 2>&1; tail -1 $O/bench_1stream.log | cut -c1-120
timeout 500 python bench.py --backbone vitl16_384 --task depth --steps 8 --warmup 3 --no-also > $O/bench_vitl16.log 2>&1; tail -1 $O/bench_vitl16.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('vitl16', d['value_inflight1'], d['value_inflight2'], d['roofline']['frac'], d['parity']['parity_mode']['value'], d['parity']['parity_mode']['max_abs'])"
timeout 300 python tools/gemm_bench.py --only cal.4096,cal.8192,vit.qkv,vit.proj,vit.fc1,vit.fc2,rcu@96,rcu@48,head.0,l2_rn,l3_rn,s2.c1,s2.c2,s2.c3 --iters 30 > $O/gemm_shapes.txt 2>&1; grep TF/s $O/gemm_shapes.txt | tail -16
timeout 200 python bench.py --gpus 1 --steps 3 --warmup 1 --dist-selftest --no-cpu-baseline --no-also --parity-dtype none --no-schedule-ab --profile-steps 1 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dist selftest:', d['config']['weight_broadcast'])" | tee $O/dist_selftest.txt
cd /tmp
export DPTX_STREAMS=1   # kernel-level passes: one launch per layer over the whole batch (the bench's per-launch figures)
B="python $R/bench.py --no-cpu-baseline --no-also --parity-dtype none --profile-steps 1 --no-schedule-ab"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- $B --steps 5 --warmup 2 > $O/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_mixed -o r -- $B --dtype mixed --steps 3 --warmup 1 > $O/trace_mixed.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o r -- $B --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o r -- $B --steps 2 --warmup 1 > $O/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS -d $O/pmc_sq -o r -- $B --steps 2 --warmup 1 > $O/pmc_sq.log 2>&1
unset DPTX_STREAMS
# two forwards in flight under the tracer as well: how much of the GPU time overlaps
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_inflight2 -o r -- python $R/bench.py --no-cpu-baseline --no-also --parity-dtype none --profile-steps 1 --no-schedule-ab --inflight 2 --steps 6 --warmup 2 > $O/trace_inflight2.log 2>&1
cd $R
db() { find $O/$1 -name "*.db" | head -1; }
python tools/rocprof_summary.py $(db trace) > $O/r06_kernel_trace_stats.txt 2>&1
python tools/rocprof_summary.py $(db trace_mixed) > $O/r06_kernel_trace_stats_mixed.txt 2>&1
python tools/rocprof_summary.py $(db trace_inflight2) > $O/r06_kernel_trace_stats_inflight2.txt 2>&1
python tools/rocprof_overlap.py $(db trace_inflight2) >> $O/r06_kernel_trace_stats_inflight2.txt 2>&1
python tools/rocprof_summary.py $(db pmc_fetch) --pmc > $O/r06_pmc_fetch_size.txt 2>&1
python tools/rocprof_summary.py $(db pmc_write) --pmc > $O/r06_pmc_write_size.txt 2>&1
python tools/rocprof_summary.py $(db pmc_sq) --pmc > $O/r06_pmc_sq.txt 2>&1
cat $O/r06_pmc_traffic.json | head -14
head -14 $O/r06_kernel_trace_stats.txt
tail -6 $O/r06_kernel_trace_stats_inflight2.txt
find $O -name "*.db" -size +20M -delete
du -sh $O
