#!/bin/bash
# configs[1] only, ten timed steps:  bash tools/ab_libs.sh tools/ab_r4e_one.sh tagA tagB ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$1]', round(d['value']/1e9,1), round(r['avg_launch_ms'],3), round(r['epilogue_kernel_ms'],3))" || tail -3 gpurun_out/ab.err
