#!/bin/bash
# epilogue time vs tiles per workgroup (PG_EPI_MIN_TILES) for a few shapes: bash tools/epi_mt.sh "128 100 36" 
cd ${GRAFT_REPO_ROOT:-/root/repo}
for MT in ${1:-128 100}; do
  python panagram_amd/build.py --force -DPG_EPI_MIN_TILES=$MT >/dev/null 2>&1
  for A in "" "--genomes 27 --genome-mb 40" "--genomes 64 --genome-mb 20 --k 31 --d 0.005" "--genomes 128 --genome-mb 10"; do
    echo -n "[MT=$MT] [$A] "
    python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-compare $A 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,1), round(d['roofline']['avg_launch_ms'],3), round(d['roofline']['epilogue_kernel_ms'],3))"
  done
done
python panagram_amd/build.py --force >/dev/null 2>&1
