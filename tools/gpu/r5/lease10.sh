#!/bin/bash
# round 5, lease 10: tile selection under the new schedule (two forwards in flight): is the 256x256 persistent kernel still the
# right default when another forward's blocks could share a CU with 128x128 tiles (64 KB of LDS, two blocks per CU)?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r5l10; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-also --parity-dtype none --steps 20 --warmup 5 --profile-steps 1 --no-schedule-ab"
for i in 1 2; do
  for V in "default" "DPTX_TILE=128" "DPTX_PP_ADV=1.0" "DPTX_PP_ADV=1.25" "DPTX_PP_ADV=2.5"; do
    if [ "$V" = "default" ]; then E=""; else E="$V"; fi
    env $E timeout 300 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V run $i:', d['value'])" | tee -a $O/ab.txt
  done
done
