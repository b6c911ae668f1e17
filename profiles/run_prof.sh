#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun):  bash profiles/run_prof.sh <tag> [bench args]
# 1) kernel trace + stats  2) PMC passes in their own runs (no trace domains mixed in)
TAG=${1:-r1}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --settle-s 0 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness --no-config5 $@"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $R/bench.py $ARGS > $OUT/trace_bench.json 2> $OUT/trace.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python $R/bench.py $ARGS > /dev/null 2> $OUT/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -o pmc -- python $R/bench.py $ARGS > /dev/null 2> $OUT/pmc_write.err
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $OUT/pmc_ea -o pmc -- python $R/bench.py $ARGS > /dev/null 2> $OUT/pmc_ea.err
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_misc -o pmc -- python $R/bench.py $ARGS > /dev/null 2> $OUT/pmc_misc.err
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -o pmc -- python $R/bench.py $ARGS > /dev/null 2> $OUT/pmc_sq.err
find $OUT -name "*.csv" | head -50
