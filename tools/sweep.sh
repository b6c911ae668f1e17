#!/bin/bash
# tuning sweep on the GPU box: rebuild the library with different slots/tile/maxrun/qcap and bench each
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=gpurun_out/sweep.txt; : > $OUT
for cfg in ${SWEEP_CFGS:-"8,512,16,192,3.0" "16,512,16,64,6.0" "16,512,16,64,5.0" "16,512,16,64,4.0" "16,1024,16,128,6.0" "16,512,12,64,6.0" "16,256,16,32,6.0"}; do
  IFS=, read SL T MR Q K <<< "$cfg"
  python panagram_amd/build.py --force -DPG_SLOTS=$SL -DPG_PROBE_TILE=$T -DPG_PROBE_MAXRUN=$MR -DPG_PROBE_QCAP=$Q 2>/dev/null
  echo "== SLOTS=$SL TILE=$T MAXRUN=$MR QCAP=$Q kpb=$K $SWEEP_ARGS" >> $OUT
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --keys-per-bucket $K $SWEEP_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value']/1e9, d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['table_bytes']/1e9)" >> $OUT
done
python panagram_amd/build.py --force 2>/dev/null
cat $OUT
