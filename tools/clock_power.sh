#!/bin/bash
# shader clock and socket power while k_probe runs back to back (is the launch time a matter of the clock the board sustains?)
cd ${GRAFT_REPO_ROOT:-/root/repo}
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | head -4
python bench.py --steps 6000 --warmup 5 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness "$@" > /tmp/cp_bench.json 2>/dev/null &
BP=$!
sleep 6
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Package Power" | tr '\n' ' '; echo; sleep 1.5; done
wait $BP
python -c "
import json; d=json.loads(open('/tmp/cp_bench.json').read().splitlines()[-1]); r=d['roofline']
print('steps', d['steps'], 'value', round(d['value']/1e9,1), 'probe ms', round(r['avg_launch_ms'],3), 'stats ms', round(r['epilogue_kernel_ms'],3))"
