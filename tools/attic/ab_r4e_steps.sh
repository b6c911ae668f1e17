#!/bin/bash
# the default shape at several step counts and with / without the legs behind the timed region (does the statistics pass's time depend on them?)
cd ${GRAFT_REPO_ROOT:-/root/repo}
F="--no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness"
for A in "--steps 10 --warmup 3 $F" "--steps 20 --warmup 5 $F" "--steps 20 --warmup 5 --no-other-shapes --no-sharded-leg --no-e2e --no-robustness" "--steps 20 --warmup 5 --no-cpu-baseline --no-other-shapes --no-sharded-leg --no-e2e --no-robustness" "--steps 20 --warmup 5"; do
  timeout 900 python bench.py $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().splitlines()[-1]); r=d['roofline']
print('[$A]', round(d['value']/1e9,1), round(r['avg_launch_ms'],3), round(r['epilogue_kernel_ms'],3))" || tail -3 gpurun_out/ab.err
done
