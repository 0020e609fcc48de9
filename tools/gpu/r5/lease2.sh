#!/bin/bash
# round 5, lease 2: residual prefetch ahead of the slab epilogue (DPTX_EPI_PREFETCH) x wave-private statistics staging
# (DPTX_STATS_WP): op tests + launch forms, then a same-box A/B matrix, 3 alternations each (bf16), 2 (mixed)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r5l2; mkdir -p $O
export TMPDIR=/tmp
python -c "from omnidata_amd.engine import load_library; print(load_library().dptx_version())" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -m gpu -q --tb=short --timeout=600 -x \
   -k "test_gpu_ops or launch_forms or deterministic" > $O/pytest.log 2>&1; echo "exit $?" >> $O/pytest.log; tail -6 $O/pytest.log
B="python bench.py --no-cpu-baseline --no-also --parity-dtype none --steps 20 --warmup 5 --profile-steps 1"
for i in 1 2 3; do
  for V in "0 0" "1 0" "0 1" "1 1"; do
    set -- $V
    DPTX_EPI_PREFETCH=$1 DPTX_STATS_WP=$2 timeout 300 $B --profile-dump $O/launches_bf16_$1$2_$i.csv > $O/ab_bf16_$1$2_$i.log 2>&1
    echo "bf16 PREFETCH=$1 STATS_WP=$2 run $i: $(tail -1 $O/ab_bf16_$1$2_$i.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_breakdown']['gemm']['ms_per_step'])")" | tee -a $O/ab.txt
  done
done
for i in 1 2; do
  for V in "0 1" "1 1"; do
    set -- $V
    DPTX_EPI_PREFETCH=$1 DPTX_STATS_WP=$2 timeout 300 $B --dtype mixed --steps 10 > $O/ab_mixed_$1$2_$i.log 2>&1
    echo "mixed PREFETCH=$1 STATS_WP=$2 run $i: $(tail -1 $O/ab_mixed_$1$2_$i.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")" | tee -a $O/ab.txt
  done
done
python - <<'PY'
import csv, glob, collections
for V in ("00", "10", "01", "11"):
    acc = collections.defaultdict(list)
    for f in sorted(glob.glob(f"gpurun_out/r5l2/launches_bf16_{V}_*.csv")):
        t = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            for k in ("attn.proj", "mlp.fc2", "attn.qkv", "resConfUnit1.conv2", "resConfUnit2.conv2", "resConfUnit1.conv1"):
                if k in r["name"]: t[k] += float(r["ms"])
        for k, v in t.items(): acc[k].append(round(v, 4))
    print("PREFETCH,STATS_WP=%s per-forward ms (single-stream profile):" % V, dict(acc))
PY
