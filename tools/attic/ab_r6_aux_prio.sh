#!/bin/bash
# the statistics stream's priority against the probe's (PG_AUX_PRIORITY) on BASELINE configs[3] / [2]-like shapes:  bash tools/ab_r6_aux_prio.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
COMMON="--no-cpu-baseline --no-other-shapes --no-sharded-leg --no-e2e --no-robustness --no-baseline-configs --no-config5 --no-compare --steps 4 --warmup 1"
for SHAPE in "--genomes 64 --genome-mb 200 --contigs 10 --k 31 --d 0.005" "--genomes 27 --genome-mb 135 --contigs 5"; do
  for P in default high low default high; do
    if [ $P = default ]; then unset PG_AUX_PRIORITY; else export PG_AUX_PRIORITY=$P; fi
    python bench.py $SHAPE $COMMON 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$SHAPE', 'aux priority $P:', round(d['value']/1e9,1), 'G, step', round(d['ms_per_step'],2), 'ms, k_probe', round(r['avg_launch_ms'],2), 'statistics', round(r['epilogue_kernel_ms'],2))"
  done
done
