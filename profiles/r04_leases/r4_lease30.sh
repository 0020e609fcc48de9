#!/bin/bash
# tile-shape sweep of the ResNetV2 convolution shapes at the batch sizes the schedule launches (32: one stream, 16: one of two)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4t; mkdir -p $O; : > $O/tile_sweep.txt
S=s0.c1,s0.c2,s0.c3,s1.c1,s1.c2,s1.c3,s2.c1,s2.c2,s2.c3,"s1.b0.c2(s2)",out_conv@96
for b in 32 16; do for t in 0 128128 12864 6464; do
  echo "== batch $b DPTX_TILE=$t" >> $O/tile_sweep.txt
  DPTX_TILE=$t timeout 120 python tools/gemm_bench.py --only "$S" --iters 50 --batch $b 2>&1 | grep "TF/s" | grep -v TOTAL >> $O/tile_sweep.txt
done; done
python - <<'P'
import re,collections
cur=None; d=collections.defaultdict(dict)
for l in open('gpurun_out/r4t/tile_sweep.txt'):
    m=re.match(r'== batch (\d+) DPTX_TILE=(\d+)',l)
    if m: cur=(m.group(1),m.group(2)); continue
    m=re.match(r'(\S+)\s+M=.*?([\d.]+) ms',l)
    if m: d[(cur[0],m.group(1))][cur[1]]=float(m.group(2))
for k in sorted(d):
    v=d[k]; best=min(v,key=v.get)
    print(k, {t:round(x*1e3,1) for t,x in v.items()}, 'best', best, 'default/best %.2f'%(v['0']/v[best]))
P
