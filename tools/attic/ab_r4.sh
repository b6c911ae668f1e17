#!/bin/bash
# round 4's shape set, one line per shape (G k-mers/s, probe ms, statistics ms):  bash tools/ab_libs.sh tools/ab_r4.sh tagA tagB ...
# PG_AB_SHAPES=narrow|two|wide|w6 picks a subset
cd ${GRAFT_REPO_ROOT:-/root/repo}
NARROW=("" "--genomes 12 --genome-mb 60" "--genomes 20 --genome-mb 40" "--genomes 27 --genome-mb 40")
TWO=("--genomes 40 --genome-mb 30" "--genomes 64 --genome-mb 20" "--genomes 64 --genome-mb 20 --k 31 --d 0.005")
WIDE=("--genomes 65 --genome-mb 10" "--genomes 128 --genome-mb 10")
W6=("--genomes 8 --genome-mb 200" "--genomes 27 --genome-mb 160" "--genomes 64 --genome-mb 160")
case "${PG_AB_SHAPES:-all}" in
  narrow) S=("${NARROW[@]}");; two) S=("${TWO[@]}");; wide) S=("${WIDE[@]}");; w6) S=("${W6[@]}");;
  *) S=("${NARROW[@]}" "${TWO[@]}" "${WIDE[@]}" "${W6[@]}");;
esac
for A in "${S[@]}"; do
  timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$1] [$A]', round(d['value']/1e9,1), round(r['avg_launch_ms'],3), round(r['epilogue_kernel_ms'],3))" || tail -3 gpurun_out/ab.err
done
