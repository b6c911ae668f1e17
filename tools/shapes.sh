cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/t.txt
for A in "" "--k 31" "--genomes 27 --genome-mb 40" "--genomes 64 --genome-mb 20 --k 31" "--k 25" "--genomes 2 --genome-mb 400"; do
  echo "== $A" >> gpurun_out/t.txt
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline $A 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value']/1e9, d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['table_bytes']/1e9)" >> gpurun_out/t.txt
done
cat gpurun_out/t.txt
