#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in "" "--genomes 64 --genome-mb 20 --k 31 --d 0.005" "--genomes 64 --genome-mb 200 --contigs 10"; do
for S in 1 2; do
for K in 1 16 64 256 1024; do
  [ "$K" == "1" ] && [ "$S" == "2" ] && continue
  PG_CHUNK_MIN_TILES=2048 PG_RUN_STREAMS=$S PG_RUN_CHUNKS=$K timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[streams=$S chunks=$K] [$A] G/s', round(d['value']/1e9,1), 'ms/step', round(d['ms_per_step'],3), 'probe span ms', round(r['avg_launch_ms'],3), 'stats span ms', round(r['epilogue_kernel_ms'],3))"
done; done; done
