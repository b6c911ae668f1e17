#!/bin/bash
# round 6: the table's density AT CREATION (PG_TABLE_KEYS_PER_LINE; the library's default is 3 keys per 128-byte line of 8 slots),
# one line per shape and density; PG_KPLS / PG_SHAPES override the lists:   bash tools/density_sweep_r6.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
IFS=';' read -ra S <<< "${PG_SHAPES:-;--genomes 27 --genome-mb 40;--genomes 64 --genome-mb 20 --contigs 10 --k 31 --d 0.005;--per-genome-launches}"
for R in 1 2; do
for K in ${PG_KPLS:-0 1.0 1.5 2.0 0}; do
  if [ "$K" = "0" ]; then unset PG_TABLE_KEYS_PER_LINE; else export PG_TABLE_KEYS_PER_LINE=$K; fi
  for A in "${S[@]}"; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness --no-config5 $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('[kpl=$K] [$A]', round(d['value']/1e9,1), 'step', round(d['ms_per_step'],3), 'probe', round(r['avg_launch_ms'],3), 'GB', round(c['table_bytes']/1e9,2), 'keys/line', c['keys_per_128B_line'], 'spill', round(c['table_spill_fraction'],3), 'm', c['minimizer_length'], 'build', round(c['table_build_s'],3))" || tail -3 gpurun_out/ab.err
  done
done
done
