#!/bin/bash
# rows of 2 to 8 bytes (9..64 genomes), one line per shape:  bash tools/ab_libs.sh tools/ab_r4e_mid.sh tagA tagB ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
S=("--genomes 12 --genome-mb 60" "--genomes 20 --genome-mb 40" "--genomes 27 --genome-mb 40" "--genomes 40 --genome-mb 30" "--genomes 64 --genome-mb 20")
for A in "${S[@]}"; do
  timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$1] [$A]', round(d['value']/1e9,1), round(r['avg_launch_ms'],3), round(r['epilogue_kernel_ms'],3))" || tail -3 gpurun_out/ab.err
done
