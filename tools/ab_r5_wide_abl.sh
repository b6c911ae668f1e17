#!/bin/bash
# round 5: the split-layout probe's timing ablations (wrong rows by design: --no-colsums skips bench.py's invariant):
#   bash tools/ab_libs.sh tools/ab_r5_wide_abl.sh sv a1 a2 a8 a3    (-DPG_ABLATE=1 every fetch a cache hit, 2 no gather and no row store,
#   8 the row store without the mask gather, 3 keys / minimizers / runs only)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in "--genomes 65 --genome-mb 10" "--genomes 96 --genome-mb 10" "--genomes 128 --genome-mb 10"; do
  timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness --no-config5 --no-colsums $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$1] [$A] probe ms', round(r['avg_launch_ms'],3))" || tail -3 gpurun_out/ab.err
done
