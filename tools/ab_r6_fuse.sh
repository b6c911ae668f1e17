#!/bin/bash
# round 6: fused statistics (PG_FUSE_STATS=1, the default) against the statistics pass over every row (=0), one line per shape and mode:
#   bash tools/ab_r6_fuse.sh [tag]          PG_SHAPES=';'-separated bench.py argument strings overrides the list
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
IFS=';' read -ra S <<< "${PG_SHAPES:---genomes 12 --genome-mb 60;--genomes 20 --genome-mb 40;--genomes 27 --genome-mb 40;--genomes 40 --genome-mb 30;--genomes 64 --genome-mb 20 --contigs 10 --k 31 --d 0.005;--genomes 65 --genome-mb 10;--genomes 80 --genome-mb 10;--genomes 96 --genome-mb 10;--genomes 112 --genome-mb 10;--genomes 128 --genome-mb 10}"
for A in "${S[@]}"; do
  for F in 0 1 0 1; do
    PG_FUSE_STATS=$F timeout 900 python bench.py --steps ${PG_STEPS:-10} --warmup 3 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness --no-config5 $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$1] [fuse=$F] [$A]', round(d['value']/1e9,1), 'step', round(d['ms_per_step'],3), 'probe', round(r['avg_launch_ms'],3), 'stats', round(r['epilogue_kernel_ms'],3))" || tail -3 gpurun_out/ab.err
  done
done
