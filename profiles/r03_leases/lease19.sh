#!/bin/bash
# round 3, GPU call 19: the launch-form test; tile-selection constants after the persistent / direct kernel (share of the chip of
# a half-batch launch, assumed per-tile advantage of the 256x256 kernel)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=$GRAFT_REPO_ROOT/gpurun_out/r3s
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q -k "launch_forms or deterministic" > $O/tests.log 2>&1; tail -2 $O/tests.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --parity-dtype none"
for rep in 1 2; do
  for CFG in "0.7 1.25" "0.5 1.25" "1.0 1.25" "0.7 1.5" "0.5 1.5" "0.7 1.1"; do
    set -- $CFG
    DPTX_CU_SHARE=$1 DPTX_PP_ADV=$2 timeout 300 $B > $O/bf16_s$1_a$2_$rep.log 2>&1; echo "share=$1 adv=$2 bf16: $(tail -1 $O/bf16_s$1_a$2_$rep.log | cut -c76-90)"
  done
done
for CFG in "0.7 1.25" "0.5 1.25" "0.7 1.5"; do
  set -- $CFG
  DPTX_CU_SHARE=$1 DPTX_PP_ADV=$2 timeout 300 $B --dtype mixed > $O/mixed_s$1_a$2.log 2>&1; echo "share=$1 adv=$2 mixed: $(tail -1 $O/mixed_s$1_a$2.log | cut -c76-90)"
done
