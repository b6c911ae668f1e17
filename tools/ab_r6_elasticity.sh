#!/bin/bash
# round 6: what extra VALU work inside k_probe costs, by row width (the budget a fused statistics pass would have):
#   bash tools/ab_libs.sh tools/ab_r6_elasticity.sh base dv16 dv32 dv48     (variants: -DPG_DUMMY_VALU=16/32/48 extra v_alignbit per batch)
cd ${GRAFT_REPO_ROOT:-/root/repo}
S=("--genomes 27 --genome-mb 40" "--genomes 64 --genome-mb 20 --contigs 10 --k 31 --d 0.005" "--genomes 65 --genome-mb 10" "--genomes 96 --genome-mb 10" "--genomes 128 --genome-mb 10" "")
for A in "${S[@]}"; do
  timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness --no-config5 $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$1] [$A]', round(d['value']/1e9,1), 'probe', round(r['avg_launch_ms'],3), 'stats', round(r['epilogue_kernel_ms'],3))" || tail -3 gpurun_out/ab.err
done
