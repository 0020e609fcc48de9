#!/bin/bash
# round 5, lease 4: bench.py with the pipelined schedule as its default -- 3 short runs (each measures BOTH schedules in one
# process: >= 3 same-box pairs), the CLI / dist tests, one full default run
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r5l4; mkdir -p $O
export TMPDIR=/tmp
python -c "from omnidata_amd.engine import load_library; print(load_library().dptx_version())" 2>&1 | tail -1
for i in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --parity-dtype none --profile-steps 1 > $O/ab_$i.log 2>&1
  tail -1 $O/ab_$i.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('run $i: inflight 2 ->', d['value'], ' inflight 1 ->', d['config']['schedule_ab']['value'])" | tee -a $O/ab.txt
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --parity-dtype none --profile-steps 1 --inflight 1 > $O/ab_inflight1.log 2>&1
tail -1 $O/ab_inflight1.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('--inflight 1 run: headline (1) ->', d['value'], ' other (2) ->', d['config']['schedule_ab']['value'])" | tee -a $O/ab.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --parity-dtype none --profile-steps 1 --inflight 3 > $O/ab_inflight3.log 2>&1
tail -1 $O/ab_inflight3.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('--inflight 3 run: headline (3) ->', d['value'])" | tee -a $O/ab.txt
timeout 900 python -m pytest tests/test_gpu_cli.py -m gpu -q --tb=short --timeout=600 > $O/pytest.log 2>&1; echo "exit $?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>$O/bench.err; tail -1 $O/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value', d['value'], 'ms', d['ms_per_step'], 'schedule_ab', d['config']['schedule_ab']['value'])
print('roofline', d['roofline']['frac'], 'parity_mode', d['parity']['parity_mode']['value'], d['parity']['parity_mode']['max_abs'], d['parity']['parity_mode']['schedule_ab'])
print('also', [(a['task'], a['dtype'], a['value']) for a in d['also']])
"; tail -3 $O/bench.err
