#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4x; mkdir -p $O
for dt in mixed bf16; do
timeout 300 python bench.py --dtype $dt --steps 10 --warmup 3 --no-cpu-baseline --no-also --parity-dtype none --profile-dump $O/launches_$dt.csv > $O/b_$dt.json 2>$O/b_$dt.err; tail -1 $O/b_$dt.json | cut -c60-200
done
