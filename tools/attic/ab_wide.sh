#!/bin/bash
# more than 64 genomes, one line per shape (value G k-mers/s, probe ms, statistics ms, table build s):  bash tools/ab_wide.sh <tag>   (PG_WIDE_LAYOUT=split: no records)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in "--genomes 65 --genome-mb 10" "--genomes 96 --genome-mb 10" "--genomes 128 --genome-mb 10" "--genomes 128 --genome-mb 10 --k 31"; do
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$1] [$A]', round(d['value']/1e9,1), round(r['avg_launch_ms'],3), round(r['epilogue_kernel_ms'],3), round(d['config']['table_build_s'],3), d['config']['table_bytes'])" || tail -3 gpurun_out/ab.err
done
