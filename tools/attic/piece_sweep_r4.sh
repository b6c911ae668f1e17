cd $GRAFT_REPO_ROOT
run() { timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness "$@" 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$*]', round(d['value']/1e9,1), round(r['avg_launch_ms'],3), round(r['epilogue_kernel_ms'],3))" || tail -3 gpurun_out/ab.err; }
for P in 1 2 4 8; do run --genomes 64 --genome-mb 40 --piece-tiles $P; done
for P in 1 2 4 8; do run --genomes 128 --genome-mb 10 --piece-tiles $P; done
for P in 4 8 16 32 64; do run --genomes 40 --genome-mb 30 --piece-tiles $P; done
for P in 4 8 16 64; do run --genomes 16 --genome-mb 60 --piece-tiles $P; done
for P in 2 4 8 64; do run --genomes 256 --genome-mb 5 --piece-tiles $P; done
