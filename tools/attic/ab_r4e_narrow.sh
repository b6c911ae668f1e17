#!/bin/bash
# one-byte rows (up to 8 genomes), by genome length and contig count:  bash tools/ab_libs.sh tools/ab_r4e_narrow.sh tagA tagB ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
S=("" "--genomes 8 --genome-mb 200" "--genomes 4 --genome-mb 100" "--genomes 8 --genome-mb 100 --contigs 2000")
for A in "${S[@]}"; do
  timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$1] [$A]', round(d['value']/1e9,1), round(r['avg_launch_ms'],3), round(r['epilogue_kernel_ms'],3))" || tail -3 gpurun_out/ab.err
done
