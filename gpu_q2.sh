#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/ksweep.py 2>&1 | grep -E "^M="
