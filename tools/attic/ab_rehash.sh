#!/bin/bash
# k_rehash / k_import_kmc / k_insert_* durations from the kernel trace of a default bench set-up:  bash tools/ab_rehash.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=/tmp/abr_$1; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg > /dev/null 2> $O/err
python - <<PY
import csv,glob
f=glob.glob("$O/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(x in r["Name"] for x in ("k_insert","k_rehash","k_import")): print("[$1]", r["Name"][:40], "calls", r["Calls"], "total ms", round(float(r["TotalDurationNs"])/1e6,2))
PY
