#!/bin/bash
# k_probe timing ablations (wrong rows by design: --no-colsums skips bench.py's invariant):
#   bash tools/ab_libs.sh tools/ab_ablate.sh base abl1 abl2 abl3 abl5     (lib_ablN.so = panagram_amd/build.py -DPG_ABLATE=N:
#   1 every fetch a cache hit, 2 no row store, 3 keys / minimizers / runs only, 5 lines fetched and staged but not scanned,
#   6 scan without the hit slot's mask read, 7 keys read from LDS but one compare instead of eight)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in "" "--genomes 64 --genome-mb 20 --k 31 --d 0.005" "--genomes 128 --genome-mb 10"; do
for i in 1 2; do
  timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-colsums $A 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$1] [$A] probe ms', round(r['avg_launch_ms'],3))"
done; done
