#!/bin/bash
# tuning sweep on the GPU box: rebuild the library with different -D knobs and bench two shapes each
#   bash tools/sweep.sh "-DPG_PROBE_TILE=256" "-DPG_PROBE_QCAP=128 -DPG_PROBE_MAXRUN=8" ...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=gpurun_out/sweep.txt; : > $OUT
for DEFS in "" "$@"; do
  python panagram_amd/build.py --force $DEFS >/dev/null 2>&1
  for A in "" "--genomes 27 --genome-mb 40"; do
    echo -n "[$DEFS] [$A] " >> $OUT
    python bench.py --steps 5 --warmup 1 --no-cpu-baseline $A 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,1), round(d['config'].get('per_genome_launches_value',0)/1e9,1))" >> $OUT 2>&1
  done
done
python panagram_amd/build.py --force >/dev/null 2>&1
cat $OUT
