#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in ${SHAPES:-"--genomes 65 --genome-mb 10" "--genomes 72 --genome-mb 10" "--genomes 80 --genome-mb 10" "--genomes 96 --genome-mb 10" "--genomes 100 --genome-mb 10" "--genomes 120 --genome-mb 10" "--genomes 128 --genome-mb 10"}; do :; done
run() {
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness $2 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; n=d['config']['positions_per_step_per_gpu']
print('[$1] [$2]', round(d['value']/1e9,1), round(r['avg_launch_ms'],3), round(r['epilogue_kernel_ms'],3), 'probe ps/pos', round(r['avg_launch_ms']*1e9/n,2), 'stats ps/pos', round(r['epilogue_kernel_ms']*1e9/n,2), 'nbytes', d['config']['nbytes'])"
}
if [ -n "$WIDE_BIG" ]; then
  for A in "--genomes 160 --genome-mb 8" "--genomes 192 --genome-mb 6" "--genomes 200 --genome-mb 6" "--genomes 256 --genome-mb 5" "--genomes 288 --genome-mb 4" "--genomes 300 --genome-mb 4" "--genomes 320 --genome-mb 4"; do run "$1" "$A"; done
else
  for A in "--genomes 65 --genome-mb 10" "--genomes 72 --genome-mb 10" "--genomes 80 --genome-mb 10" "--genomes 96 --genome-mb 10" "--genomes 100 --genome-mb 10" "--genomes 120 --genome-mb 10" "--genomes 128 --genome-mb 10"; do run "$1" "$A"; done
fi
