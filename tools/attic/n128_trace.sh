cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for A in "--genomes 128 --genome-mb 10" "--genomes 96 --genome-mb 20"; do
O=$R/gpurun_out/kt_n; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-compare $A > $O/bench.json 2> $O/err
echo "== $A"; python -c "import sys,json; d=json.loads(open('$O/bench.json').read()); print(d['value']/1e9, d['ms_per_step'])"
python - <<PY
import csv,glob
f=glob.glob("$O/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r['Name'][:90], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'])
PY
done
