#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in "" "--genomes 16 --genome-mb 50" "--genomes 27 --genome-mb 40" "--genomes 40 --genome-mb 30" "--genomes 64 --genome-mb 20" "--genomes 65 --genome-mb 10" "--genomes 128 --genome-mb 10"; do
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$1] [$A]', round(d['value']/1e9,1), round(r['avg_launch_ms'],3), round(r['epilogue_kernel_ms'],3))"
done
