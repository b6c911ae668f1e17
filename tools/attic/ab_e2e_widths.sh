#!/bin/bash
# the files-to-files leg (Index.run()) by genome count: seconds, and the anchor + write part as GB of rows per second
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in "--genomes 8 --genome-mb 40" "--genomes 12 --genome-mb 40" "--genomes 20 --genome-mb 40" "--genomes 27 --genome-mb 40" "--genomes 40 --genome-mb 40" "--genomes 56 --genome-mb 40" "--genomes 64 --genome-mb 40" "--genomes 72 --genome-mb 20" "--genomes 100 --genome-mb 20" "--genomes 128 --genome-mb 20"; do
  PG_BENCH_E2E_ANY=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-robustness $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); e=d['e2e']; nb=d['config']['nbytes']
rows=e['positions']*nb
print('[$A] e2e s', round(e['seconds'],3), 'read+parse+sketch', round(e['read_parse_sketch_s'],3), 'insert', round(e['table_insert_s'],3), 'anchor+write', round(e['anchor_and_write_s'],3), '=', round(rows/e['anchor_and_write_s']/1e9,2), 'GB rows/s;', round(e['anchor_and_write_s']*1e12/e['positions'],1), 'ps/pos; out MB', round(e['index_bytes_out']/1e6), 'in MB', round(e['fasta_bytes_in']/1e6))"
done
