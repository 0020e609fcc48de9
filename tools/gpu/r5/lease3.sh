#!/bin/bash
# round 5, lease 3: ForwardPipeline (several forwards in flight, shared weights): tests, the probe in the parity mode, and the
# driver's default bench command with the new schedule
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r5l3; mkdir -p $O
export TMPDIR=/tmp
python -c "from omnidata_amd.engine import load_library; print(load_library().dptx_version())" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_cli.py -m gpu -q --tb=short --timeout=600 > $O/pytest.log 2>&1; echo "exit $?" >> $O/pytest.log; tail -12 $O/pytest.log
timeout 400 python tools/gpu/r5/pipeline_probe.py --dtype mixed --steps 10 2>&1 | grep img/s | head -9 | tee $O/pipeline_probe_mixed.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>$O/bench.err; tail -1 $O/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value', d['value'], 'ms', d['ms_per_step'], 'schedule_ab', d['config']['schedule_ab'])
print('roofline', d['roofline']['frac'], 'parity', json.dumps(d['parity'])[:900])
print('also', [(a['task'], a['dtype'], a['value']) for a in d['also']])
print('cpu', d['cpu_baseline'])
"; tail -3 $O/bench.err
