#!/bin/bash
# round 3, GPU call 6: GEMM tile selection aware of the stream split (cu_share) -- validation + A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3f
mkdir -p $O
export TMPDIR=/tmp
python -c "from omnidata_amd.engine import load_library; print(load_library().dptx_version())" > $O/version.log 2>&1; cat $O/version.log
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_stress.py tests/test_gpu_dual.py tests/test_gpu_mixed.py -m gpu -q --tb=short --timeout=900 --maxfail=8 > $O/pytest_gpu.log 2>&1; echo "exit $?" >> $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log
line() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also > $O/bench_share_$rep.log 2>&1; line $O/bench_share_$rep.log share0.5_$rep
DPTX_CU_SHARE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also > $O/bench_share1_$rep.log 2>&1; line $O/bench_share1_$rep.log share1_$rep
done
DPTX_CU_SHARE=0.7 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also > $O/bench_share07.log 2>&1; line $O/bench_share07.log share0.7
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --dtype mixed > $O/bench_mixed.log 2>&1; line $O/bench_mixed.log mixed_share0.5
DPTX_CU_SHARE=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --dtype mixed > $O/bench_mixed1.log 2>&1; line $O/bench_mixed1.log mixed_share1
du -sh $O
