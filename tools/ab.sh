#!/bin/bash
# A/B of two build variants on ONE box (box-to-box spread is ~2 %): bash tools/ab.sh "<defs A>" "<defs B>" [rounds]
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in $(seq 1 ${3:-3}); do
  for V in "$1" "$2"; do
    python panagram_amd/build.py --force $V >/dev/null 2>&1
    for A in "" "--k 31"; do
      echo -n "[$V] [$A] "
      python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-compare $A 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,1), round(d['roofline']['avg_launch_ms'],3))"
    done
  done
done
python panagram_amd/build.py --force >/dev/null 2>&1
