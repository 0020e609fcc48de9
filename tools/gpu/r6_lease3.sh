#!/bin/bash
# round 6, lease 3: fp8 ViT (DPTX_FLAG_FP8_VIT): tests, op-level rates against the bf16 kernels, the bench line's `also` entries
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r6l3; mkdir -p $O
export TMPDIR=/tmp
python -c "from omnidata_amd.engine import load_library; print(load_library().dptx_version())" > $O/version.log 2>&1; cat $O/version.log
timeout 1500 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_cli.py -m gpu -q -s --timeout=900 > $O/pytest_fp8.log 2>&1; tail -6 $O/pytest_fp8.log; grep "deg" $O/pytest_fp8.log | head -20
timeout 300 python tools/gemm_bench.py --dtype fp8 --only vit.qkv,vit.fc1,vit.fc2,rcu@96 --iters 30 2>&1 | grep "TF/s" | tee $O/gemm_fp8_vit.txt
timeout 300 python tools/gemm_bench.py --only vit.qkv,vit.fc1,vit.fc2,rcu@96 --iters 30 2>&1 | grep "TF/s" | tee -a $O/gemm_fp8_vit.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --parity-dtype none > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6l3/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "value_inflight1", "value_inflight2")})
for a in d["also"]:
    print(a["task"], a["dtype"], a["value_inflight1"], a["value_inflight2"], a.get("verdict"))
PY
tail -3 $O/bench.err
