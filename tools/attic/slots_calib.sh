#!/bin/bash
# calibration of the 8- vs 16-slot line choice: spill fraction and throughput, both widths forced
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in "" "--genomes 27 --genome-mb 40" "--genomes 40 --genome-mb 30" "--genomes 40 --genome-mb 30 --d 0.003" "--genomes 64 --genome-mb 20 --k 31" "--genomes 64 --genome-mb 20 --k 31 --d 0.005" "--genomes 64 --genome-mb 20 --k 21 --d 0.005" "--genomes 16 --genome-mb 50 --d 0.02"; do
  for S in 8 16; do
    PG_TABLE_SLOTS=$S python bench.py --steps 5 --warmup 1 --no-cpu-baseline $A 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; print('$A', 'slots', c['table_slots_per_line'], 'spill', round(c['table_spill_fraction'],3), 'cosched', round(d['value']/1e9,1), 'per-genome', round(c['per_genome_launches_value']/1e9,1))"
  done
done
