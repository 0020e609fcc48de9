#!/bin/bash
# round 5, lease 6: launch-form switches under the NEW schedule (two forwards in flight): persistent blocks per XCD, tile selection share
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r5l6; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-also --parity-dtype none --steps 20 --warmup 5 --profile-steps 1 --no-schedule-ab"
for i in 1 2; do
  for V in "32 0" "16 0" "0 0" "32 0.5" "16 0.5" "24 0"; do
    set -- $V
    DPTX_PERSIST=$1 DPTX_CU_SHARE_WHOLE=$2 timeout 300 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('PERSIST=$1 CU_SHARE_WHOLE=$2 run $i:', d['value'])" | tee -a $O/ab.txt
  done
done
for i in 1 2; do
  for V in "32" "16" "0"; do
    DPTX_PERSIST=$V timeout 300 $B --dtype mixed --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mixed PERSIST=$V run $i:', d['value'])" | tee -a $O/ab.txt
  done
done
