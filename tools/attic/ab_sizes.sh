#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in "--genomes 8 --genome-mb 100" "--genomes 8 --genome-mb 140" "--genomes 8 --genome-mb 200" "--genomes 8 --genome-mb 500" "--genomes 8 --genome-mb 700" "--genomes 8 --genome-mb 1500" "--genomes 4 --genome-mb 100" "--genomes 2 --genome-mb 400" "--genomes 8 --genome-mb 100 --k 31" "--genomes 8 --genome-mb 100 --k 15" "--genomes 8 --genome-mb 100 --k 25" "--genomes 8 --genome-mb 100 --k 32"; do
  timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; n=d['config']['positions_per_step_per_gpu']; c=d['config']
print('[$1] [$A]', round(d['value']/1e9,1), 'step', round(d['ms_per_step'],3), 'probe', round(r['avg_launch_ms'],3), 'stats', round(r['epilogue_kernel_ms'],3), 'probe ps/pos', round(r['avg_launch_ms']*1e9/n,2), 'm', c.get('minimizer'), 'table GB', round(c['table_bytes']/1e9,1), 'keys/line', c['keys_per_128B_line'])"
done
