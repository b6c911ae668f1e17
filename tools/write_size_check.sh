cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS="--steps 2 --warmup 1 --settle-s 0 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness --no-config5 --no-colsums"
for T in cur a3n; do
  if [ $T = a3n ]; then cp $R/panagram_amd/libpanagram_hip.so /tmp/lib_orig.so; cp $R/build_variants/lib_a3n.so $R/panagram_amd/libpanagram_hip.so; fi
  rm -rf /tmp/ws_$T; rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d /tmp/ws_$T -o pmc -- python $R/bench.py $ARGS > /dev/null 2> /tmp/ws_$T.err
  python - <<PY
import pandas as pd
df = pd.read_csv("/tmp/ws_$T/pmc_counter_collection.csv")
df = df[df.Kernel_Name.str.contains("k_probe|k_epilogue")]
df["K"] = df.Kernel_Name.str.replace("void ","").str.slice(0,30)
print("$T"); print(df.groupby(["K","Counter_Name"]).Counter_Value.mean().unstack().to_string())
PY
done
cp /tmp/lib_orig.so $R/panagram_amd/libpanagram_hip.so
