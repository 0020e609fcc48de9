mkdir -p gpurun_out/r4u
timeout 900 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_poison.py -m gpu -q --tb=short -s -k "fp8" > gpurun_out/r4u/fp8_tests.log 2>&1
grep -v "^$" gpurun_out/r4u/fp8_tests.log | grep "\[default\]\|\[trained\]\|passed\|failed\|Error\|assert" | head -20
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r4u/bench_default.json 2>gpurun_out/r4u/bench_default.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r4u/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity'])
for a in d['also']: print(a['task'], a['dtype'], a['value'], a.get('max_abs_vs_oracle'))
print(d['cpu_baseline'])
P
