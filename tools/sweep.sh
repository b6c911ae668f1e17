#!/bin/bash
# tuning sweep on the GPU box: rebuild the library with different knobs and bench each
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=gpurun_out/sweep.txt; : > $OUT
for cfg in ${SWEEP_CFGS:-"512,16,192,2,3.0" "512,16,192,1,3.0" "512,16,192,3,3.0" "512,16,192,2,2.0" "512,16,128,2,2.0" "512,16,192,2,2.5" "1024,16,384,2,3.0" "512,20,192,2,3.0"}; do
  IFS=, read T MR Q LV K <<< "$cfg"
  python panagram_amd/build.py --force -DPG_PROBE_TILE=$T -DPG_PROBE_MAXRUN=$MR -DPG_PROBE_QCAP=$Q -DPG_PROBE_STAGED_LEVELS=$LV 2>/dev/null
  echo "== TILE=$T MAXRUN=$MR QCAP=$Q LEVELS=$LV kpb=$K $SWEEP_ARGS" >> $OUT
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --keys-per-bucket $K $SWEEP_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value']/1e9, d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['table_bytes']/1e9)" >> $OUT
done
python panagram_amd/build.py --force 2>/dev/null
cat $OUT
