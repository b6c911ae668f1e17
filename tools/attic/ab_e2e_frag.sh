#!/bin/bash
# the files-to-files leg on fragmented assemblies (many contigs: 100 bins of text per contig)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in "--genomes 8 --genome-mb 100 --contigs 5" "--genomes 8 --genome-mb 100 --contigs 2000" "--genomes 8 --genome-mb 100 --contigs 20000" "--genomes 27 --genome-mb 40 --contigs 4000"; do
  PG_BENCH_E2E_ANY=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-robustness $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); e=d['e2e']
print('[$A] e2e s', round(e['seconds'],3), 'read+parse+sketch', round(e['read_parse_sketch_s'],3), 'insert', round(e['table_insert_s'],3), 'anchor+write', round(e['anchor_and_write_s'],3), 'out MB', round(e['index_bytes_out']/1e6), 'in MB', round(e['fasta_bytes_in']/1e6))"
done
