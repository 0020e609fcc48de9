mkdir -p gpurun_out/r4v
for ns in 2 3 2 3; do
for dt in bf16 mixed; do
DPTX_STREAMS=$ns timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --dtype $dt > gpurun_out/r4v/b_${dt}_$ns.json 2>/dev/null
python - <<P
import json
d=json.loads(open('gpurun_out/r4v/b_${dt}_$ns.json').read().strip().splitlines()[-1])
print('streams $ns', '$dt', d['value'], d['ms_per_step'])
P
done
done
