#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for NR in 0 5 50; do
for A in "" "--genomes 27 --genome-mb 40" "--genomes 64 --genome-mb 20"; do
  PG_BENCH_N_RUNS=$NR timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; n=d['config']['positions_per_step_per_gpu']
print('[N runs per contig: $NR] [$A]', round(d['value']/1e9,1), 'step', round(d['ms_per_step'],3), 'probe', round(r['avg_launch_ms'],3), 'stats', round(r['epilogue_kernel_ms'],3), 'probe ps/pos', round(r['avg_launch_ms']*1e9/n,2))"
done
done
