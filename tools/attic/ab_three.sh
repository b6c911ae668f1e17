#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in "" "--k 31" "--genomes 27 --genome-mb 40" "--genomes 64 --genome-mb 20"; do
  timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$1] [$A]', round(d['value']/1e9,1), round(r['avg_launch_ms'],3))"
done
