#!/bin/bash
# round 3, GPU call 1: validation of the cleaned GEMM dispatch, the LayerNorm fold and the new attention kernel + A/B lines
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3a
mkdir -p $O
export TMPDIR=/tmp
python -c "from omnidata_amd.engine import load_library; print(load_library().dptx_version())" > $O/version.log 2>&1; cat $O/version.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short --timeout=900 -x > $O/pytest_gpu.log 2>&1; echo "exit $?" >> $O/pytest_gpu.log; tail -15 $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 --profile-dump $O/launches.csv > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-400
line() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'], d['kernel_breakdown'])"; }
DPTX_LN_FOLD=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --profile-dump $O/launches_nofold.csv > $O/bench_nofold.log 2>&1; line $O/bench_nofold.log nofold
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --dtype fp16 > $O/bench_fp16.log 2>&1; line $O/bench_fp16.log fp16
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --dtype mixed > $O/bench_mixed.log 2>&1; line $O/bench_mixed.log mixed
DPTX_LN_FOLD=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --dtype mixed > $O/bench_mixed_nofold.log 2>&1; line $O/bench_mixed_nofold.log mixed_nofold
DPTX_STREAMS=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also > $O/bench_1stream.log 2>&1; line $O/bench_1stream.log 1stream
timeout 300 python tools/gemm_bench.py --only vit.qkv,vit.proj,vit.fc1,vit.fc2,rcu@96,head.0,s2.c1,s2.c2,s2.c3,cal.4096 --iters 30 > $O/gemm_shapes.txt 2>&1; grep TF/s $O/gemm_shapes.txt
cd /tmp
export DPTX_STREAMS=1
B="python $R/bench.py --no-cpu-baseline --no-also --parity-dtype none --profile-steps 1"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- $B --steps 5 --warmup 2 > $O/trace.log 2>&1
unset DPTX_STREAMS
cd $R
db() { find $O/$1 -name "*.db" | head -1; }
python tools/rocprof_summary.py $(db trace) > $O/r03a_kernel_trace_stats.txt 2>&1
head -14 $O/r03a_kernel_trace_stats.txt
find $O -name "*.db" -size +20M -delete
du -sh $O
