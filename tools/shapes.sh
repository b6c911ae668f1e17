#!/bin/bash
# bench.py over a few pangenome shapes on the GPU box (one line each: G k-mers/s co-scheduled,
# G k-mers/s with one launch per genome, k_probe ms per launch, roofline frac, table GB)
cd ${GRAFT_REPO_ROOT:-/root/repo}
: > gpurun_out/t.txt
for A in "" "--k 31" "--genomes 27 --genome-mb 40" "--genomes 64 --genome-mb 20 --k 31 --d 0.005" "--k 25" "--genomes 2 --genome-mb 400" "$@"; do
  echo "== $A" >> gpurun_out/t.txt
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline $A 2>gpurun_out/shapes.err | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(round(d['value']/1e9,1), round(d['config'].get('per_genome_launches_value',0)/1e9,1), round(d['roofline']['avg_launch_ms'],3), d['roofline']['frac'], round(d['config']['table_bytes']/1e9,1))" >> gpurun_out/t.txt 2>&1
done
cat gpurun_out/t.txt
