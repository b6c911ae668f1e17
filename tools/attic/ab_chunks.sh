#!/bin/bash
# a whole run as K chunks (probe of chunk c+1 beside the statistics of chunk c) against one probe launch + one pass (K=1)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in "" "--genomes 27 --genome-mb 40" "--genomes 64 --genome-mb 20 --k 31 --d 0.005" "--genomes 128 --genome-mb 10" "${EXTRA:-}"; do
[ "$A" == "" ] && [ "$DONE_DEFAULT" == "1" ] && continue
DONE_DEFAULT=1
for K in 1 4 8 16 32; do
  PG_RUN_CHUNKS=$K timeout 900 python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[chunks=$K] [$A] G/s', round(d['value']/1e9,1), 'ms/step', round(d['ms_per_step'],3), 'probe span ms', round(r['avg_launch_ms'],3), 'stats span ms', round(r['epilogue_kernel_ms'],3))"
done; done
