#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_prepost.py tests/test_gpu_cli.py -m gpu -q --tb=short --timeout=600 2>&1 | tail -15
