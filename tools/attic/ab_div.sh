#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in "--d 0.001" "--d 0.01" "--d 0.03" "--d 0.05" "--d 0.1" "--genomes 27 --genome-mb 40 --d 0.05" "--genomes 64 --genome-mb 20 --d 0.03"; do
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; n=d['config']['positions_per_step_per_gpu']; c=d['config']
print('[$A]', round(d['value']/1e9,1), 'step', round(d['ms_per_step'],3), 'probe', round(r['avg_launch_ms'],3), 'stats', round(r['epilogue_kernel_ms'],3), 'probe ps/pos', round(r['avg_launch_ms']*1e9/n,2), 'keys', c['table_keys'], 'table GB', round(c['table_bytes']/1e9,1), 'keys/line', c['keys_per_128B_line'], 'build s', round(c['table_build_s'],3))"
done
