cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
BR_L=200000000 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/df -o t -- python $R/tools/bgzf_rate.py 2>&1 | tail -3
python - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/df/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print(r['Name'][:60], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
