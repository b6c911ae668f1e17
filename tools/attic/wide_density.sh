#!/bin/bash
# bench.py at more than 64 genomes vs table density (keys per 128-byte key line; --no-rehash: the library's own sizing)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in "--genomes 128 --genome-mb 10" "--genomes 256 --genome-mb 5"; do
for D in "--keys-per-bucket 2" "--keys-per-bucket 3" "--keys-per-bucket 4" "--keys-per-bucket 6" "--no-rehash"; do
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg $A $D 2>gpurun_out/wide.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('$A $D |', round(d['value']/1e9,1), 'G/s probe', round(r['avg_launch_ms'],2), 'stats', round(r['epilogue_kernel_ms'],2), 'table GB', round(c['table_bytes']/1e9,1), 'spill', round(c['table_spill_fraction'],3))"
done; done
