#!/bin/bash
# staged lines per step vs narrow-window shapes:  bash tools/ab_libs.sh tools/ab_maxrun.sh base mr24 mr32
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in "--genomes 64 --genome-mb 20 --minimizer 18" "--genomes 64 --genome-mb 20 --minimizer 16" "--genomes 8 --genome-mb 100 --minimizer 18" "--genomes 64 --genome-mb 200 --contigs 10 --no-rehash" ""; do
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$1] [$A]', round(d['value']/1e9,1), round(r['avg_launch_ms'],3), round(r['epilogue_kernel_ms'],3))"
done
