#!/bin/bash
# minimizer window (w = k - m + 1) by genome count, k=21:  bash tools/w_sweep.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in "--genomes 12 --genome-mb 60" "--genomes 16 --genome-mb 50" "--genomes 27 --genome-mb 40" "--genomes 40 --genome-mb 30" "--genomes 64 --genome-mb 20"; do
for M in 15 16 17 18; do
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg $A --minimizer $M 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('$A m=$M w=$((21-M+1)) |', round(d['value']/1e9,1), 'probe', round(r['avg_launch_ms'],2), 'stats', round(r['epilogue_kernel_ms'],2), 'spill', round(c['table_spill_fraction'],3))"
done; done
