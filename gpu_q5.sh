#!/bin/bash
cd "$GRAFT_REPO_ROOT"
SH="vit.qkv,vit.proj,vit.fc1,vit.fc2,rcu@96,rcu@48,head.0,l2_rn,head.2,s2.c2"
for v in 0 1 2 3; do echo "== DPTX_V=$v"; DPTX_V=$v timeout 200 python tools/gemm_bench.py --only $SH 2>&1 | grep -E "TOTAL|vit.fc1|rcu@96|vit.qkv|head.0"; done
