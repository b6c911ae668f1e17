#!/bin/bash
# one rocprofv3 --pmc pass with the given counters over a bench run; per-kernel means:  bash tools/pmc_set.sh "<counters>" [bench args]
CNT=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=/tmp/pmcset; rm -rf $OUT; mkdir -p $OUT
timeout ${PMC_TIMEOUT:-150} rocprofv3 --kernel-include-regex "k_probe|k_epilogue|k_insert_tile" --pmc $CNT --output-format csv -d $OUT -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg "$@" > /dev/null 2> $OUT/err || tail -5 $OUT/err
python - <<PY
import pandas as pd, glob
pd.set_option("display.width", 250); pd.set_option("display.max_columns", 30)
f = glob.glob("$OUT/**/pmc_counter_collection.csv", recursive=True)
df = pd.read_csv(f[0])
df = df[df.Kernel_Name.str.contains("k_probe|k_epilogue|k_insert_tile")]
df["K"] = df.Kernel_Name.str.replace("void ","").str.slice(0,40)
print(df.groupby(["K","Counter_Name"]).Counter_Value.mean().unstack().T.to_string())
PY
