#!/bin/bash
# 8- vs 16-slot lines on re-hashed tables, wide window:  bash tools/slots_calib2.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in "--genomes 27 --genome-mb 40" "--genomes 64 --genome-mb 20" "--genomes 64 --genome-mb 20 --k 31 --d 0.005"; do
  for S in 8 16; do for KPB in 2 3; do
    PG_TABLE_SLOTS=$S timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --keys-per-bucket $KPB $A 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; print('$A', 'slots', c['table_slots_per_line'], 'kpb $KPB spill', round(c['table_spill_fraction'],3), 'G/s', round(d['value']/1e9,1), 'probe', round(r['avg_launch_ms'],3))"
  done; done
done
