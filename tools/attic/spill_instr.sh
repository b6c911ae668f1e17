#!/bin/bash
# how much of k_probe's instruction count is the overflow path?  VALU wave-instructions and launch time of the
# default bench at several table densities (fewer keys per line = fewer keys outside their home line)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for KPB in ${@:-0.5 1 2 3}; do
  OUT=$R/gpurun_out/spill_$KPB
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d $OUT -o pmc -- \
    python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --keys-per-bucket $KPB > $OUT/bench.json 2> $OUT/err.txt
  python - <<PY
import pandas as pd, json
df = pd.read_csv("$OUT/pmc_counter_collection.csv")
a = df[df.Kernel_Name.str.contains("k_probe")]
m = a.groupby("Counter_Name").Counter_Value.mean()
d = json.loads(open("$OUT/bench.json").read().splitlines()[-1])
pos = d["config"]["positions_per_step_per_gpu"]
print("kpb=$KPB spill=%.3f value=%.1fG launch=%.3fms  VALU/pos=%.3f LDS/pos=%.3f SALU/pos=%.3f VMEMRD/pos=%.3f wait_any=%.2f active=%.2f" % (
    d["config"]["table_spill_fraction"], d["value"]/1e9, d["roofline"]["avg_launch_ms"], m["SQ_INSTS_VALU"]/pos, m["SQ_INSTS_LDS"]/pos,
    m["SQ_INSTS_SALU"]/pos, m["SQ_INSTS_VMEM_RD"]/pos, m["SQ_WAIT_ANY"]/m["SQ_WAVE_CYCLES"], m["SQ_ACTIVE_INST_ANY"]/m["SQ_WAVE_CYCLES"]))
PY
done
