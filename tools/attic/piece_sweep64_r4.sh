cd $GRAFT_REPO_ROOT
run() { timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness "$@" 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$*]', round(d['value']/1e9,1), round(d['ms_per_step'],2), round(r['avg_launch_ms'],3), round(r['epilogue_kernel_ms'],3))" || tail -3 gpurun_out/ab.err; }
for P in 64 8 16 64 8; do run --genomes 64 --genome-mb 160 --piece-tiles $P; done
for P in 64 8 64 8; do run --genomes 64 --genome-mb 20 --piece-tiles $P; done
for P in 64 8 64 8; do run --genomes 64 --genome-mb 20 --k 31 --d 0.005 --piece-tiles $P; done
