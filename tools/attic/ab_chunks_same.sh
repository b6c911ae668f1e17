#!/bin/bash
# the statistics of a chunk right behind its probe on ONE stream (PG_CHUNK_SAME_STREAM=1), K chunks: does the pass read its rows out of the Infinity Cache?
cd ${GRAFT_REPO_ROOT:-/root/repo}
for K in 1 2 4 8 16; do
  for A in "" "--genomes 27 --genome-mb 40" "--genomes 64 --genome-mb 20"; do
  PG_RUN_CHUNKS=$K PG_CHUNK_SAME_STREAM=1 PG_CHUNK_MIN_TILES=1024 timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[K=$K] [$A]', round(d['value']/1e9,1), round(d['ms_per_step'],3), round(r['avg_launch_ms'],3), round(r['epilogue_kernel_ms'],3))" || tail -3 gpurun_out/ab.err
  done
done
