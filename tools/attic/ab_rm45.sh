#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in "--genomes 8 --blocks 8" "--genomes 16 --genome-mb 50 --blocks 2" "--genomes 64 --genome-mb 20 --blocks 8" "--genomes 64 --genome-mb 20 --blocks 2" "--genomes 128 --genome-mb 10 --blocks 8"; do
timeout 600 python bench.py --mode genome-sharded --steps 4 --warmup 1 --no-cpu-baseline --no-other-shapes --no-e2e --no-robustness $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']
print('[$1] [$A]', round(d['value']/1e9,1), 'G k-mers/s; ms/step', round(d['ms_per_step'],3))"
done
