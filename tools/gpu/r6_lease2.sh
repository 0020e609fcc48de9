#!/bin/bash
# round 6, lease 2: new tests (measured schedule, overlap probe, eval harness), the full bench line with telemetry, in-run traffic
# measurement, A/B of GEMM translation units built without packed fp32 arithmetic
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r6l2; mkdir -p $O
export TMPDIR=/tmp
python -c "from omnidata_amd.engine import load_library; print(load_library().dptx_version())" > $O/version.log 2>&1; cat $O/version.log
ls /sys/class/drm > $O/sysfs.txt 2>&1; python - >> $O/sysfs.txt 2>&1 <<'PY'
import torch
from omnidata_amd.telemetry import GpuTelemetry, device_pci_bus_id, find_card
print("pci", device_pci_bus_id(0), "card", find_card(device_pci_bus_id(0)))
t = GpuTelemetry(0); print(t.files); print(t.read_once())
PY
cat $O/sysfs.txt | tail -4
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_eval_checkpoint.py "tests/test_gpu_e2e.py::test_measured_schedule_choice_is_bit_identical_and_reported" "tests/test_gpu_e2e.py::test_two_stream_split_is_bit_identical" -m gpu -x -q -s --timeout=900 > $O/pytest_new.log 2>&1; tail -5 $O/pytest_new.log; grep -i "probe:\|measured schedule" $O/pytest_new.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6l2/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "value_inflight1", "value_inflight2", "ms_per_step", "headline_schedule")})
print("telemetry", json.dumps(d["telemetry"])[:900])
print("schedule", d["config"]["engine_schedule"], "roofline", d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_source"])
print("parity", {k: (v.get("value_inflight1"), v.get("value_inflight2"), v.get("max_abs")) for k, v in d["parity"].items() if isinstance(v, dict) and "value" in v})
print("also", [(a["task"], a["dtype"], a["value_inflight1"], a["value_inflight2"], a.get("verdict")) for a in d["also"]])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["threads_tried_images_per_s"])
PY
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-also --parity-dtype none --no-schedule-ab --measure-traffic --traffic-out $O/traffic.json > $O/bench_traffic.json 2> $O/bench_traffic.err; cat $O/traffic.json; tail -3 $O/bench_traffic.err
for rep in 1 2 3; do
  for lib in "" _nopk; do
    DPTX_LIB=$PWD/omnidata_amd/libdptx$lib.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --parity-dtype none --profile-steps 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep lib[$lib]', d['value_inflight1'], d['value_inflight2'], d['config']['engine_schedule'], d['telemetry'].get('inflight1',{}).get('sclk_mhz'))" | tee -a $O/nopk_ab.txt
  done
done
for lib in "" _nopk; do
  echo "== lib[$lib]" >> $O/nopk_gemm.txt
  DPTX_LIB=$PWD/omnidata_amd/libdptx$lib.so timeout 300 python tools/gemm_bench.py --only vit.qkv,vit.proj,vit.fc1,vit.fc2,rcu@96,head.0,s2.c3 --iters 30 2>&1 | grep "TF/s" >> $O/nopk_gemm.txt
done
cat $O/nopk_gemm.txt
