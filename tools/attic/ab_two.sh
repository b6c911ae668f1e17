#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in "" "--k 31" "--genomes 128 --genome-mb 10" "--genomes 64 --genome-mb 20 --k 31 --d 0.005" "--genomes 27 --genome-mb 40"; do
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-shapes --no-sharded-leg $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$1] [$A]', round(d['value']/1e9,1), round(r['avg_launch_ms'],3), round(r['epilogue_kernel_ms'],3), 'per-genome', round(d['config'].get('per_genome_launches_value',0)/1e9,1))"
done
