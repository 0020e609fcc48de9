#!/bin/bash
# round 2, first lease: full -m gpu suite on HEAD, stale-read probe, precision frontier, bench line with parity leg
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2a
mkdir -p $O
export TMPDIR=/tmp
rocminfo | grep -E "Marketing|Compute Unit|Max Clock" | head -6 > $O/rocminfo.txt 2>&1
( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/stale_probe tools/gpu/stale_probe.hip && timeout 300 /tmp/stale_probe 20000 ) > $O/stale_probe.log 2>&1; tail -8 $O/stale_probe.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short --timeout=900 -s > $O/pytest_gpu.log 2>&1; echo "exit $?" >> $O/pytest_gpu.log; grep -E "passed|failed|exit|FAILED|Error" $O/pytest_gpu.log | tail -15
timeout 600 python tools/precision_frontier.py --steps 10 --out $O/frontier.md > $O/frontier.log 2>&1; tail -20 $O/frontier.log
timeout 600 python bench.py --steps 20 --warmup 5 --profile-dump $O/launches.csv > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-600
du -sh $O
