#!/bin/bash
# SQ counters of every pg:: kernel of a bench run:  bash tools/pmc_kernels.sh <tag> [bench args]
TAG=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmck_$TAG
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 2 --warmup 1 --settle-s 0 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness --no-config5 $@"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/a -o pmc -- python $R/bench.py $ARGS > $OUT/bench.json 2> $OUT/a.err
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -o pmc -- python $R/bench.py $ARGS > /dev/null 2> $OUT/b.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o trace -- python $R/bench.py $ARGS > /dev/null 2> $OUT/t.err
python - <<PY
import pandas as pd
pd.set_option("display.width", 250); pd.set_option("display.max_columns", 30)
frames=[]
for d in ("a","b"):
    df = pd.read_csv("$OUT/%s/pmc_counter_collection.csv" % d)
    df = df[df.Kernel_Name.str.contains("k_probe|k_epilogue|k_insert|k_cols")]
    df["K"] = df.Kernel_Name.str.replace("void ","").str.slice(0,34)
    frames.append(df.groupby(["K","Counter_Name"]).Counter_Value.mean().unstack())
m = pd.concat(frames, axis=1)
print(m.T.to_string())
ks = pd.read_csv("$OUT/t/trace_kernel_stats.csv"); ks["Name"]=ks["Name"].str.slice(0,60)
print(ks.head(6).to_string(index=False))
PY
