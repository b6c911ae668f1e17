#!/bin/bash
# quick SQ counter pass:  bash profiles/run_pmc_sq.sh <tag> [bench args]
TAG=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--steps 2 --warmup 1 --settle-s 0 --no-cpu-baseline --no-e2e --no-robustness --no-config5 $@"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -o pmc -- python $R/bench.py $ARGS > $OUT/bench.json 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_FLAT --output-format csv -d $OUT/pmc_sq2 -o pmc -- python $R/bench.py $ARGS > /dev/null 2> $OUT/pmc_sq2.err
python - <<PY
import pandas as pd
for d in ("pmc_sq","pmc_sq2"):
    try:
        df = pd.read_csv("$OUT/%s/pmc_counter_collection.csv" % d)
    except Exception as e:
        print(d, "missing", e); continue
    a = df[df.Kernel_Name.str.contains("k_probe")]
    print(a.groupby("Counter_Name").Counter_Value.mean().to_string())
PY
cat $OUT/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(d['value']/1e9, d['roofline']['avg_launch_ms'])"
