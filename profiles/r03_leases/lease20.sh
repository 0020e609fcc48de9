#!/bin/bash
# round 3, GPU call 20: tile-selection constants, second matrix (more launches to the persistent / direct 256x256 kernel)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=$GRAFT_REPO_ROOT/gpurun_out/r3t
mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --parity-dtype none"
for rep in 1 2; do
  for CFG in "0.5 1.5" "0.5 2.0" "0.5 3.0" "0.4 1.5" "0.4 2.0" "0.3 2.0" "0.35 1.75" "0.5 1.75"; do
    set -- $CFG
    DPTX_CU_SHARE=$1 DPTX_PP_ADV=$2 timeout 300 $B > $O/bf16_s$1_a$2_$rep.log 2>&1; echo "share=$1 adv=$2 bf16: $(tail -1 $O/bf16_s$1_a$2_$rep.log | cut -c76-90)"
  done
done
for CFG in "0.5 1.5" "0.5 2.0" "0.4 2.0"; do
  set -- $CFG
  DPTX_CU_SHARE=$1 DPTX_PP_ADV=$2 timeout 300 $B --dtype mixed > $O/mixed_s$1_a$2.log 2>&1; echo "share=$1 adv=$2 mixed: $(tail -1 $O/mixed_s$1_a$2.log | cut -c76-90)"
done
