#!/bin/bash
# kernel-trace a few bench shapes on the GPU box:  bash tools/ktrace.sh  (args sets below)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for A in "" "--genomes 27 --genome-mb 40" "--genomes 64 --genome-mb 20 --k 31"; do
  O=$R/gpurun_out/kt_$i; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $A > $O/bench.json 2> $O/err
  echo "== $A"; cat $O/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value']/1e9)"
  python - <<PY
import csv,glob
f=glob.glob("$O/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r['Name'][:70], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'])
PY
  i=$((i+1))
done
