#!/bin/bash
# tuning sweep on the GPU box: rebuild the library with different wg/tile/lines/unroll and bench each
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=gpurun_out/sweep.txt; : > $OUT
for cfg in ${SWEEP_CFGS:-"256,1024,320,4,3.0" "256,1024,320,8,3.0" "256,2048,640,4,3.0" "128,512,160,4,3.0" "128,1024,320,4,3.0" "256,512,160,4,3.0" "256,1024,320,4,2.0" "256,1024,384,4,4.0"}; do
  IFS=, read WG T LN U K <<< "$cfg"
  python panagram_amd/build.py --force -DPG_ANCHOR_WG=$WG -DPG_ANCHOR_TILE=$T -DPG_ANCHOR_LINES=$LN -DPG_ANCHOR_UNROLL=$U 2>/dev/null
  echo "== WG=$WG TILE=$T LINES=$LN UNROLL=$U kpb=$K $SWEEP_ARGS" >> $OUT
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --keys-per-bucket $K $SWEEP_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value']/1e9, d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['table_bytes']/1e9)" >> $OUT
done
python panagram_amd/build.py --force 2>/dev/null
cat $OUT
