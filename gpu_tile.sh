#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
SH="vit.qkv,vit.fc1,vit.fc2,rcu@96,head.0,l2_rn,rcu@24"
for t in 0 12864 6464; do echo "== DPTX_TILE=$t"; DPTX_TILE=$t timeout 200 python tools/gemm_bench.py --only $SH 2>&1 | grep -E "TF/s"; done
