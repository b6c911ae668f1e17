#!/bin/bash
# the files-to-files leg of bench.py, three runs: seconds, read+parse+sketch, inserts, anchor+write   bash tools/ab_libs.sh tools/e2e_line.sh tagA tagB
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-robustness 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); e=d['e2e']; print('[$1]', round(e['seconds'],4), round(e['read_parse_sketch_s'],4), round(e['table_insert_s'],4), round(e['anchor_and_write_s'],4))" || tail -5 gpurun_out/ab.err
done
