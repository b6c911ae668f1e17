#!/bin/bash
# round 6: tiles per wave of k_probe (PG_PROBE_TPW; the next tile's descriptors and bases prefetched during the overflow drain):
#   bash tools/ab_libs.sh tools/ab_r6_tpw.sh base tpw base tpw     (base: the commit before, one tile per wave; tpw: this tree)
cd ${GRAFT_REPO_ROOT:-/root/repo}
S=("" "--genomes 27 --genome-mb 40" "--genomes 64 --genome-mb 20 --contigs 10 --k 31 --d 0.005" "--genomes 65 --genome-mb 10" "--genomes 128 --genome-mb 10" "--per-genome-launches")
TPWS="${PG_TPWS:-0}"
[ "$1" != "base" ] && TPWS="${PG_TPWS:-1 2 4 8}"
for A in "${S[@]}"; do
  for T in $TPWS; do
    if [ "$T" = "0" ]; then unset PG_PROBE_TPW; else export PG_PROBE_TPW=$T; fi
    timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness --no-config5 $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$1] [tpw=$T] [$A]', round(d['value']/1e9,1), 'step', round(d['ms_per_step'],3), 'probe', round(r['avg_launch_ms'],3), 'stats', round(r['epilogue_kernel_ms'],3))" || tail -3 gpurun_out/ab.err
  done
done
