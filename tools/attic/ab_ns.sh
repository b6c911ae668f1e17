#!/bin/bash
# the north star's shape (w = 6 instantiations) at a fifth and at full size
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in "--genomes 64 --genome-mb 40 --contigs 2" "--genomes 64 --genome-mb 200 --contigs 10"; do
  timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$1] [$A]', round(d['value']/1e9,1), round(r['avg_launch_ms'],3), round(r['epilogue_kernel_ms'],3))"
done
