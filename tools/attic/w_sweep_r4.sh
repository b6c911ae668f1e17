#!/bin/bash
# round 4b: the minimizer length again, on the kernel whose batches are cut at 16 runs (G k-mers/s, k_probe ms, statistics ms, keys per line)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness "$@" 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$*]', round(d['value']/1e9,1), round(r['avg_launch_ms'],3), round(r['epilogue_kernel_ms'],3), 'keys/line', d['config']['keys_per_128B_line'])" || tail -3 gpurun_out/ab.err; }
for S in ${PG_SWEEP_SHAPES:-"--genomes 4 --genome-mb 700" "--genomes 8 --genome-mb 100 --k 20" "--genomes 8 --genome-mb 100 --k 22" "--genomes 8 --genome-mb 100 --k 24"}; do
  for M in ${PG_SWEEP_M:-15 16 17 18}; do run $S --minimizer $M; done
done
