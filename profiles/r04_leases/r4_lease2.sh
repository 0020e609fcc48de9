mkdir -p gpurun_out/r4b
timeout 400 python tools/gpu/r4_join.py 200 > gpurun_out/r4b/join.log 2>&1
tail -14 gpurun_out/r4b/join.log
for i in 1 2 3; do timeout 200 python -m pytest tests/test_gpu_fp8.py -m gpu -q -k "dual_task_runs" --tb=line 2>&1 | tail -2; done > gpurun_out/r4b/fp8dual_x3.log 2>&1
tail -6 gpurun_out/r4b/fp8dual_x3.log
timeout 900 python -m pytest tests -m gpu -q --tb=short --timeout=600 > gpurun_out/r4b/pytest_full.log 2>&1
tail -8 gpurun_out/r4b/pytest_full.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --dtype mixed --profile-dump gpurun_out/r4b/mixed_launches.csv > gpurun_out/r4b/bench_mixed.json 2>gpurun_out/r4b/bench_mixed.err
tail -c 600 gpurun_out/r4b/bench_mixed.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --profile-dump gpurun_out/r4b/bf16_launches.csv > gpurun_out/r4b/bench_bf16.json 2>gpurun_out/r4b/bench_bf16.err
tail -c 600 gpurun_out/r4b/bench_bf16.json
