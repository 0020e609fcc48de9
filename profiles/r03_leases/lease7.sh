#!/bin/bash
# round 3, GPU call 7: which share of the chip should a half-batch GEMM be tiled for?  (256x256 rule x narrow-tile thresholds)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3g
mkdir -p $O
export TMPDIR=/tmp
line() { tail -1 $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
for pp in 1 0.7 0.5; do for sm in 1 0.5; do
DPTX_CU_SHARE=$pp DPTX_CU_SHARE_SMALL=$sm timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also > $O/b_${pp}_${sm}_$rep.log 2>&1; line $O/b_${pp}_${sm}_$rep.log "bf16 pp=$pp small=$sm rep$rep"
done; done; done
for pp in 1 0.7 0.5; do for sm in 1 0.5; do
DPTX_CU_SHARE=$pp DPTX_CU_SHARE_SMALL=$sm timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --dtype mixed > $O/m_${pp}_${sm}.log 2>&1; line $O/m_${pp}_${sm}.log "mixed pp=$pp small=$sm"
done; done
du -sh $O
