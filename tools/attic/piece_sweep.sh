#!/bin/bash
# co-scheduling granularity (tiles per piece) on many-genome shapes:  bash tools/piece_sweep.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in "--genomes 64 --genome-mb 20 --k 31 --d 0.005" "--genomes 27 --genome-mb 40" ""; do
for P in 0 1 2 4 8 16; do
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg $A --piece-tiles $P 2>gpurun_out/ps.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$A] piece-tiles $P |', round(d['value']/1e9,1), round(r['avg_launch_ms'],3), round(r['epilogue_kernel_ms'],3))"
done; done
