#!/bin/bash
# the default shape only, three runs:  bash tools/ab_libs.sh tools/ab_default.sh tagA tagB
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do
  timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$1]', round(d['value']/1e9,1), round(r['avg_launch_ms'],3), round(r['epilogue_kernel_ms'],3))"
done
