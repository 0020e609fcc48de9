#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_x3.py -m gpu -q --tb=short --timeout=400 -k "attention" 2>&1 | tail -3
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['kernel_breakdown'])"
