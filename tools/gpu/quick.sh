#!/bin/bash
# quick GPU regression: op tests, e2e, x3, bench
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -m gpu -q --tb=short --timeout=600 2>&1 | tail -6
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['kernel_breakdown'])"
