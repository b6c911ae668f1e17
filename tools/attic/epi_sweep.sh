#!/bin/bash
# epilogue grid sweep on the GPU box: PG_EPI_MIN_TILES x shapes through tools/split_time.py
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for MT in ${EPI_MT:-8 16 32 64}; do
  python panagram_amd/build.py --force -DPG_EPI_MIN_TILES=$MT 2>/dev/null >/dev/null
  for A in "" "--genomes 27 --genome-mb 40" "--genomes 64 --genome-mb 20 --k 31"; do
    echo "== MIN_TILES=$MT $A"; timeout 300 python tools/split_time.py $A 2>&1 | grep -E "positions|rror"
  done
done
python panagram_amd/build.py --force 2>/dev/null >/dev/null
