#!/bin/bash
# tuning sweep on the GPU box: rebuild the library with different tile/unroll and bench each
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=gpurun_out/sweep.txt; : > $OUT
for cfg in "2048 8" "1024 8" "1024 16" "2048 16" "1024 4" "512 8"; do
  set -- $cfg
  python panagram_amd/build.py --force -DPG_ANCHOR_TILE=$1 -DPG_ANCHOR_UNROLL=$2 2>/dev/null
  for kpb in 2.0; do
    echo "== TILE=$1 UNROLL=$2 kpb=$kpb" >> $OUT
    python bench.py --steps 5 --warmup 1 --no-cpu-baseline --keys-per-bucket $kpb 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value']/1e9, d['roofline']['avg_launch_ms'], d['roofline']['frac'])" >> $OUT
  done
done
python panagram_amd/build.py --force 2>/dev/null
cat $OUT
