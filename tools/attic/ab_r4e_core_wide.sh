#!/bin/bash
# as ab_r4e_core.sh, the split-layout shapes only (k_epilogue_chunks):  bash tools/ab_libs.sh tools/ab_r4e_core_wide.sh tagA tagB ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
S=("--genomes 65 --genome-mb 10" "--genomes 128 --genome-mb 10" "--genomes 200 --genome-mb 5")
for A in "${S[@]}"; do
 for D in 0.01 0.0001; do
  timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness $A --d $D 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$1] [$A --d $D]', round(d['value']/1e9,1), round(r['avg_launch_ms'],3), round(r['epilogue_kernel_ms'],3))" || tail -3 gpurun_out/ab.err
 done
done
