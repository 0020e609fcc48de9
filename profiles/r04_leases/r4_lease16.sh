mkdir -p gpurun_out/r4p
timeout 300 python -m pytest tests/test_gpu_mixed.py -m gpu -q --tb=short -x -k "fused_head_tail" > gpurun_out/r4p/x3head_tests.log 2>&1
tail -3 gpurun_out/r4p/x3head_tests.log
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --dtype mixed --profile-dump gpurun_out/r4p/mixed_launches.csv > gpurun_out/r4p/bench_mixed.json 2>gpurun_out/r4p/bench_mixed.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r4p/bench_mixed.json').read().strip().splitlines()[-1])
print('mixed', d['value'], d['ms_per_step'])
P
done
grep "head\|output_conv" gpurun_out/r4p/mixed_launches.csv
