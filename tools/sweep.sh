#!/bin/bash
# tuning sweep on the GPU box: rebuild the library with different knobs and bench each
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=gpurun_out/sweep.txt; : > $OUT
for cfg in ${SWEEP_CFGS:-"-DPG_PROBE_NB=1,2.0" "-DPG_W6,2.0" "-DPG_W6,3.0" "-DPG_W6,1.5" "-DPG_W6:-DPG_PROBE_QCAP=128,2.0"}; do
  IFS=, read DEFS K <<< "$cfg"
  python panagram_amd/build.py --force ${DEFS//:/ } 2>/dev/null
  echo "== $DEFS kpb=$K $SWEEP_ARGS" >> $OUT
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --keys-per-bucket $K $SWEEP_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value']/1e9, d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['table_bytes']/1e9)" >> $OUT
done
python panagram_amd/build.py --force 2>/dev/null
cat $OUT
