#!/bin/bash
# tuning sweep on the GPU box: rebuild the library with different -D knobs and bench each
# SWEEP_CFGS="defs,kpb ..."  (defs separated by ':'), SWEEP_ARGS = extra bench.py arguments
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=gpurun_out/sweep.txt; : > $OUT
for cfg in ${SWEEP_CFGS:-"-DPG_PROBE_QCAP=192,2.0" "-DPG_PROBE_QCAP=128,2.0" "-DPG_PROBE_TILE=256,2.0" "-DPG_PROBE_STAGED_LEVELS=1,2.0"}; do
  IFS=, read DEFS K <<< "$cfg"
  python panagram_amd/build.py --force ${DEFS//:/ } 2>/dev/null
  for A in "${SWEEP_ARGSETS[@]:-}"; do
    echo "== $DEFS kpb=$K $A" >> $OUT
    python bench.py --steps 5 --warmup 1 --no-cpu-baseline --keys-per-bucket $K $A 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value']/1e9, d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['table_bytes']/1e9)" >> $OUT
  done
done
python panagram_amd/build.py --force 2>/dev/null
cat $OUT
