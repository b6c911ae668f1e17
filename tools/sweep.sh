#!/bin/bash
# tuning sweep on the GPU box: rebuild the library with different knobs and bench each
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=gpurun_out/sweep.txt; : > $OUT
for cfg in ${SWEEP_CFGS:-"1,512,16,192,2.0" "2,512,16,192,2.0" "2,1024,16,320,2.0" "2,512,16,160,2.0" "2,768,16,256,2.0"}; do
  IFS=, read NB T MR Q K <<< "$cfg"
  python panagram_amd/build.py --force -DPG_PROBE_NB=$NB -DPG_PROBE_TILE=$T -DPG_PROBE_MAXRUN=$MR -DPG_PROBE_QCAP=$Q 2>/dev/null
  echo "== NB=$NB TILE=$T MAXRUN=$MR QCAP=$Q kpb=$K $SWEEP_ARGS" >> $OUT
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --keys-per-bucket $K $SWEEP_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value']/1e9, d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['table_bytes']/1e9)" >> $OUT
done
python panagram_amd/build.py --force 2>/dev/null
cat $OUT
