#!/bin/bash
# the chunk-parallel statistics pass by chunks per row: 65 / 128 genomes (one 16-byte chunk), 200 / 256 (two), 300 (three):  bash tools/ab_libs.sh tools/ab_r4e_wide2.sh tagA tagB ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
S=("--genomes 65 --genome-mb 10" "--genomes 128 --genome-mb 10" "--genomes 200 --genome-mb 5" "--genomes 256 --genome-mb 5" "--genomes 300 --genome-mb 4")
for A in "${S[@]}"; do
  timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$1] [$A]', round(d['value']/1e9,1), round(r['avg_launch_ms'],3), round(r['epilogue_kernel_ms'],3))" || tail -3 gpurun_out/ab.err
done
