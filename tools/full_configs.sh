#!/bin/bash
# BASELINE configs 3 and 4 and the human-scale shape at FULL size on one GPU (the bench line's own timing, further legs off)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in "--genomes 27 --genome-mb 135" "--genomes 64 --genome-mb 200 --contigs 10 --k 31 --d 0.005" "--genomes 8 --genome-mb 3000 --contigs 24 --d 0.001"; do
  timeout 1200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness $A 2>gpurun_out/full.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; n=d['config']['positions_per_step_per_gpu']; c=d['config']
print('[$A]', round(d['value']/1e9,1), 'G k-mers/s; step ms', round(d['ms_per_step'],2), 'probe', round(r['avg_launch_ms'],2), 'stats', round(r['epilogue_kernel_ms'],2), 'positions', n, 'keys', c['table_keys'], 'table GB', round(c['table_bytes']/1e9,1), 'build s', round(c['table_build_s'],2), 'frac', round(r['frac'],3))"
done
