#!/bin/bash
# round 6, lease 1: sanity of the tree (new ADVICE tests + e2e), image-group A/B, deep-pipeline GEMM variants on the deep-K shapes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r6l1; mkdir -p $O
export TMPDIR=/tmp
python -c "from omnidata_amd.engine import load_library; print(load_library().dptx_version())" > $O/version.log 2>&1; cat $O/version.log
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_e2e.py -m gpu -x -q --timeout=600 > $O/pytest_subset.log 2>&1; tail -3 $O/pytest_subset.log
timeout 900 python tools/gpu/r6_groups_ab.py --dtype bf16 --reps 3 > $O/groups_bf16.txt 2>&1; tail -14 $O/groups_bf16.txt
timeout 600 python tools/gpu/r6_groups_ab.py --dtype mixed --reps 2 --groups 1,4 > $O/groups_mixed.txt 2>&1; tail -8 $O/groups_mixed.txt
SH="s2.c1,s2.c2,s1.c2,l3_rn,rcu@24,rcu@12,pp4.conv2,l2_rn,rcu@48,patch.proj"
for t in 0 3128128 4128128 2256128 3256128 2128256 3128256; do
  echo "== DPTX_TILE=$t" >> $O/gemm_ns.txt
  DPTX_TILE=$t timeout 300 python tools/gemm_bench.py --only $SH --iters 30 2>&1 | grep "TF/s" >> $O/gemm_ns.txt
done
cat $O/gemm_ns.txt
