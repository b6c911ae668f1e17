#!/bin/bash
# minimizer length at FULL-SIZE shapes (is 4^m >= 4 x keys too careful for many-genome pangenomes?):  bash tools/m_sweep_full.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { M=$1; shift; timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-rehash "$@" --minimizer $M 2>gpurun_out/msf.err | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
    print('$* m=$M |', round(d['value']/1e9,1), 'G/s | probe', round(r['avg_launch_ms'],2), 'stats', round(r['epilogue_kernel_ms'],2), 'ms | keys', c['table_keys'], 'build s', round(c['table_build_s'],3), 'spill', round(c['table_spill_fraction'],3))
except Exception as e: print('$* m=$M | failed', e)"; }
for M in 15 16 17 18; do run $M --genomes 64 --genome-mb 200 --contigs 10; done
for M in 15 16 18; do run $M --genomes 27 --genome-mb 135; done
for M in 16 17 18; do run $M --genomes 8 --genome-mb 3000 --contigs 24 --d 0.001; done
