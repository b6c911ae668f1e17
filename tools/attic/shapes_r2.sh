#!/bin/bash
# round-2 refresh of DESIGN §4's "other shapes": value co-scheduled / one launch per genome, probe and statistics ms, build s
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/shapes_r2.txt; : > $OUT
run() { timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-shapes --no-sharded-leg "$@" 2>gpurun_out/shapes.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('$*', '|', round(d['value']/1e9,1), '/', round(c.get('per_genome_launches_value',0)/1e9,1), 'G/s | probe', round(r['avg_launch_ms'],2), 'stats', round(r['epilogue_kernel_ms'],2), 'ms | keys', c['table_keys'], 'table GB', round(c['table_bytes']/1e9,1), 'build s', round(c['table_build_s'],3))" >> $OUT 2>&1; }
run
run --k 31
run --k 25
run --genomes 2 --genome-mb 400
run --genomes 16 --genome-mb 50
run --genomes 27 --genome-mb 40
run --genomes 40 --genome-mb 30
run --genomes 64 --genome-mb 20
run --genomes 64 --genome-mb 20 --k 31 --d 0.005
run --genomes 4 --genome-mb 100 --contigs 20000
run --genomes 27 --genome-mb 135 --no-rehash --no-compare
run --genomes 64 --genome-mb 200 --k 31 --d 0.005 --contigs 10 --no-rehash --no-compare
run --genomes 8 --genome-mb 3000 --contigs 24 --d 0.001 --no-rehash --no-compare
cat $OUT
