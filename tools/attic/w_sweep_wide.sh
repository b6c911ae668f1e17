#!/bin/bash
# minimizer window by shape (default: the split layout, > 64 genomes):  [SHAPES="--genomes 64 --genome-mb 20;..."] bash tools/w_sweep_wide.sh [k=21]
cd ${GRAFT_REPO_ROOT:-/root/repo}
K=${1:-21}
IFS=";" read -ra LIST <<< "${SHAPES:---genomes 128 --genome-mb 10;--genomes 96 --genome-mb 10;--genomes 256 --genome-mb 5}"
for A in "${LIST[@]}"; do
for W in 8 7 6 5 4; do
  M=$((K-W+1))
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --k $K $A --minimizer $M 2>gpurun_out/wsw.err | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
    print('k=$K $A m=$M w=$W |', round(d['value']/1e9,1), 'probe', round(r['avg_launch_ms'],2), 'stats', round(r['epilogue_kernel_ms'],2), 'spill', round(c['table_spill_fraction'],3), 'keys/line', c['keys_per_128B_line'])
except Exception as e: print('k=$K $A m=$M w=$W | failed', e)"
done; done
