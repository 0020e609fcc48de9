#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4y; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_mixed.py -q -k "two_mfma or per_layer or default_model_is or mixed_b32 or batch_32 or B32" -s 2>&1 | tail -15 > $O/tests.log; cat $O/tests.log
timeout 300 python -m pytest tests/test_gpu_poison.py tests/test_gpu_dual.py -q -k "mixed or default" 2>&1 | tail -3
for v in 2 3; do
DPTX_HEAD0_MFMAS=$v timeout 300 python bench.py --dtype mixed --steps 10 --warmup 3 --no-cpu-baseline --no-also > $O/b_mixed_h$v.json 2>$O/b_mixed_h$v.err
python - $O/b_mixed_h$v.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], d.get('parity'))
P
done
timeout 300 python bench.py --dtype mixed --steps 10 --warmup 3 --no-cpu-baseline --no-also --parity-dtype none --profile-dump $O/launches_mixed.csv > /dev/null 2>&1
grep -E "output_conv.0|fusion.up|head.tail" $O/launches_mixed.csv | tail -4
