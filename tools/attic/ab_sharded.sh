#!/bin/bash
# genome-sharded bench mode on one GPU playing rank 0 of 8 (one-genome block table): columns straight from the probe or via rows
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { timeout 300 python bench.py --mode genome-sharded --blocks 8 --steps 5 --warmup 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('[$1]', round(d['value']/1e9,1), 'ms', round(d['ms_per_step'],3), 'table GB', round(c['block_table_bytes']/1e9,2), 'direct', c['columns_from_the_probe'])"; }
for i in 1 2 3; do
run "direct"
PG_COLUMNS_DIRECT=0 run "rows"
done
