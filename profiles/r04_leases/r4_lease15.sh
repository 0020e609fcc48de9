mkdir -p gpurun_out/r4o
timeout 600 python -m pytest tests/test_gpu_mixed.py -m gpu -q --tb=short -x -s -k "fused_head_tail or default_model_is or b32_vs_oracle or policy_end_to_end" > gpurun_out/r4o/x3head_tests.log 2>&1
tail -25 gpurun_out/r4o/x3head_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --dtype mixed --profile-dump gpurun_out/r4o/mixed_launches.csv > gpurun_out/r4o/bench_mixed.json 2>gpurun_out/r4o/bench_mixed.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r4o/bench_mixed.json').read().strip().splitlines()[-1])
print('mixed', d['value'], d['ms_per_step'], d['kernel_breakdown'])
P
DPTX_HEAD_FUSED_X3=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --dtype mixed > gpurun_out/r4o/bench_mixed_unfused.json 2>/dev/null
python - <<'P'
import json
d=json.loads(open('gpurun_out/r4o/bench_mixed_unfused.json').read().strip().splitlines()[-1])
print('mixed unfused head', d['value'], d['ms_per_step'])
P
grep "head\|output_conv" gpurun_out/r4o/mixed_launches.csv
