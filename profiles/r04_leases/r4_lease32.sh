#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4u2; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_x3.py tests/test_gpu_vitl16.py tests/test_gpu_flex.py -q -x -k "attention or vitl16 or flex" 2>&1 | tail -3
timeout 200 python tools/gpu/att_probe.py 2>&1 | grep -v amdgpu.ids | head -8
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --profile-dump $O/launches.csv > $O/b.json 2>$O/err.txt; python - <<'P'
import json,csv
d=json.loads(open('gpurun_out/r4u2/b.json').read().strip().splitlines()[-1]); print('bf16', d['value'], d['ms_per_step'], d['parity'])
r=[float(x['ms']) for x in csv.DictReader(open('gpurun_out/r4u2/launches.csv')) if x['name']=='attention']; print('attention per launch', sum(r)/len(r))
P
