#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in "--genomes 16 --genome-mb 40" "--genomes 27 --genome-mb 40" "--genomes 64 --genome-mb 20" "--genomes 128 --genome-mb 10"; do
 for C in "" "--no-colsums"; do
  timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness $A $C 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; n=d['config']['positions_per_step_per_gpu']
print('[$1] [$A $C]', round(d['value']/1e9,1), 'step ms', round(d['ms_per_step'],3), 'probe', round(r['avg_launch_ms'],3), 'stats', round(r['epilogue_kernel_ms'],3), 'stats ps/pos', round(r['epilogue_kernel_ms']*1e9/n,2), 'GB/s', round(n*d['config']['nbytes']/r['epilogue_kernel_ms']/1e6))"
 done
done
