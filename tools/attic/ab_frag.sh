#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in "--genomes 8 --genome-mb 100 --contigs 5" "--genomes 8 --genome-mb 100 --contigs 2000" "--genomes 8 --genome-mb 100 --contigs 20000" "--genomes 27 --genome-mb 40 --contigs 5" "--genomes 27 --genome-mb 40 --contigs 4000" "--genomes 64 --genome-mb 20 --contigs 2000"; do
  timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; n=d['config']['positions_per_step_per_gpu']
print('[$1] [$A]', round(d['value']/1e9,1), 'step ms', round(d['ms_per_step'],3), 'probe', round(r['avg_launch_ms'],3), 'stats', round(r['epilogue_kernel_ms'],3), 'probe ps/pos', round(r['avg_launch_ms']*1e9/n,2))"
done
