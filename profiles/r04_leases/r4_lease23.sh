mkdir -p gpurun_out/r4w
timeout 120 python -m pytest tests/test_gpu_mixed.py -m gpu -q --tb=short -x -k "fused_head_tail" --timeout=60 > gpurun_out/r4w/x3head_tests.log 2>&1
tail -3 gpurun_out/r4w/x3head_tests.log
for d in 0 8 1 2; do DPTX_HX_DBG=$d timeout 60 python tools/gpu/r4_headx3_bench.py 2>&1 | grep DBG; done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --dtype mixed > gpurun_out/r4w/bench_mixed.json 2>gpurun_out/r4w/bench_mixed.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r4w/bench_mixed.json').read().strip().splitlines()[-1])
print('mixed', d['value'], d['ms_per_step'])
P
