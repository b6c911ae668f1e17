#!/bin/bash
# non-temporal row stores at full size:  bash tools/ab_libs.sh tools/ab_nt.sh base nt base nt
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in "--genomes 64 --genome-mb 200 --k 31 --d 0.005 --contigs 10 --no-rehash" "--genomes 64 --genome-mb 200 --contigs 10 --no-rehash" "--genomes 64 --genome-mb 20" "--genomes 128 --genome-mb 10" "--genomes 57 --genome-mb 20" "--genomes 256 --genome-mb 5"; do
  timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$1] [$A]', round(d['value']/1e9,1), round(r['avg_launch_ms'],3), round(r['epilogue_kernel_ms'],3))"
done
