#!/bin/bash
# bench.py on pangenomes of more than 64 genomes (rows wider than 8 bytes): value, k_probe and
# statistics-kernel time per launch.   bash tools/wide_shapes.sh [tag]
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/wide_${1:-x}.txt
: > $OUT
for A in "--genomes 128 --genome-mb 10" "--genomes 96 --genome-mb 10" "--genomes 65 --genome-mb 10" "--genomes 160 --genome-mb 8" "--genomes 256 --genome-mb 5" "--genomes 300 --genome-mb 4"; do
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg $A 2>gpurun_out/wide.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$A', 'G/s', round(d['value']/1e9,1), 'probe ms', round(r['avg_launch_ms'],2), 'stats ms', round(r['epilogue_kernel_ms'],2), 'positions', d['config']['positions_per_step_per_gpu'], 'build s', round(d['config']['table_build_s'],3))" >> $OUT 2>&1
done
cat $OUT
