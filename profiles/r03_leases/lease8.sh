#!/bin/bash
# round 3, GPU call 8: per-launch times at B = 16 vs B = 32 on ONE stream (what does the half-batch split cost per layer?)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3h
mkdir -p $O
export TMPDIR=/tmp
export DPTX_STREAMS=1
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --parity-dtype none --batch 32 --profile-dump $O/launches_b32.csv > $O/b32.log 2>&1; tail -1 $O/b32.log | cut -c1-150
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --parity-dtype none --batch 16 --profile-dump $O/launches_b16.csv > $O/b16.log 2>&1; tail -1 $O/b16.log | cut -c1-150
DPTX_CU_SHARE=0.7 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --parity-dtype none --batch 16 --profile-dump $O/launches_b16_s07.csv > $O/b16s.log 2>&1; tail -1 $O/b16s.log | cut -c1-150
