#!/bin/bash
# the genome-sharded pipeline on ONE GPU (rank 0 of a layout with B blocks of G/B genomes each): per-rank rate by block width
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in "--genomes 8 --blocks 1" "--genomes 8 --blocks 2" "--genomes 8 --blocks 4" "--genomes 8 --blocks 8" "--genomes 16 --genome-mb 50 --blocks 2" "--genomes 16 --genome-mb 50 --blocks 16" "--genomes 24 --genome-mb 40 --blocks 8" "--genomes 64 --genome-mb 20 --blocks 8" "--genomes 64 --genome-mb 20 --blocks 2"; do
timeout 600 python bench.py --mode genome-sharded --steps 4 --warmup 1 --no-cpu-baseline --no-other-shapes --no-e2e --no-robustness $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']
print('[$A]', round(d['value']/1e9,1), 'G k-mers/s; ms/step', round(d['ms_per_step'],3), 'block table GB', round(c.get('block_table_bytes',0)/1e9,2), 'direct columns', c.get('columns_from_the_probe'), 'positions', c.get('positions_per_step_per_gpu'))"
done
