#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4z; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_x3.py tests/test_gpu_mixed.py -q -x -k "not b32 and not B32 and not overflow" 2>&1 | tail -4
for v in 0 1 0 1; do
DPTX_PP2=$v timeout 300 python bench.py --dtype mixed --steps 10 --warmup 3 --no-cpu-baseline --no-also --parity-dtype none > $O/b_mixed_pp2_$v.json 2>$O/err.txt
python - $O/b_mixed_pp2_$v.json $v <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('DPTX_PP2='+sys.argv[2], d['value'], d['ms_per_step'])
P
done
DPTX_PP2=1 timeout 300 python bench.py --dtype mixed --steps 10 --warmup 3 --no-cpu-baseline --no-also --parity-dtype none --profile-dump $O/launches_pp2.csv > /dev/null 2>&1
DPTX_PP2=0 timeout 300 python bench.py --dtype mixed --steps 10 --warmup 3 --no-cpu-baseline --no-also --parity-dtype none --profile-dump $O/launches_lock.csv > /dev/null 2>&1
