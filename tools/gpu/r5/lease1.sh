#!/bin/bash
# round 5, lease 1: new tests (co-residency regression + positive control, launch forms incl. the wave-private statistics
# epilogue, bench self-spawn, RCCL self-test) and a same-box A/B of the statistics epilogue (3 alternations, bf16 + mixed)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r5l1; mkdir -p $O
export TMPDIR=/tmp
python -c "from omnidata_amd.engine import load_library; print(load_library().dptx_version())" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_coresidency.py tests/test_gpu_cli.py tests/test_gpu_e2e.py -m gpu -q --tb=short --timeout=600 \
   -k "coresidency or positive_control or launch_forms or spawns_two or selftest or share_one_gpu" > $O/pytest.log 2>&1; echo "exit $?" >> $O/pytest.log; tail -15 $O/pytest.log
B="python bench.py --no-cpu-baseline --no-also --parity-dtype none --steps 20 --warmup 5 --profile-steps 1"
for i in 1 2 3; do
  for V in 0 1; do
    DPTX_STATS_WP=$V timeout 300 $B --profile-dump $O/launches_bf16_${V}_$i.csv > $O/ab_bf16_${V}_$i.log 2>&1
    echo "bf16 STATS_WP=$V run $i: $(tail -1 $O/ab_bf16_${V}_$i.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_breakdown']['gemm']['ms_per_step'])")" | tee -a $O/ab.txt
  done
done
for i in 1 2; do
  for V in 0 1; do
    DPTX_STATS_WP=$V timeout 300 $B --dtype mixed --steps 10 > $O/ab_mixed_${V}_$i.log 2>&1
    echo "mixed STATS_WP=$V run $i: $(tail -1 $O/ab_mixed_${V}_$i.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")" | tee -a $O/ab.txt
  done
done
timeout 200 python tools/gemm_bench.py --only vit.proj,vit.fc2,vit.qkv,vit.fc1 --iters 30 > $O/gemm_shapes.txt 2>&1; grep TF/s $O/gemm_shapes.txt
python - <<'PY'
import csv, glob, collections
for V in (0, 1):
    acc = collections.defaultdict(list)
    for f in sorted(glob.glob(f"gpurun_out/r5l1/launches_bf16_{V}_*.csv")):
        t = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            for k in ("attn.proj", "mlp.fc2", "attn.qkv", "mlp.fc1", "patch_embed.proj"):
                if k in r["name"]: t[k] += float(r["ms"])
        for k, v in t.items(): acc[k].append(round(v, 4))
    print("STATS_WP=%d per-forward ms (single-stream profile):" % V, dict(acc))
PY
