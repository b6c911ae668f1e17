#!/bin/bash
# repeat-family content vs unique sequence:  bash tools/ab_libs.sh tools/ab_repeats.sh base gc2
cd ${GRAFT_REPO_ROOT:-/root/repo}
for ARGS in "--genomes 27 --copies 2000 --background-mb 20" "--genomes 27 --copies 2000" "--genomes 2 --copies 2000 --background-mb 20" "--genomes 27 --copies 1 --elem 100 --background-mb 27"; do
  for M in -1 18; do echo -n "[$1] [$ARGS] minimizer $M: "; timeout 600 python tools/repeat_stress.py $ARGS --minimizer $M 2>&1 | grep "^anchor"; done
done
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('[$1] default bench', round(d['value']/1e9,1), round(d['roofline']['avg_launch_ms'],3))"
