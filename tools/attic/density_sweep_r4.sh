cd $GRAFT_REPO_ROOT
run() { timeout 900 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-e2e --no-robustness "$@" 2>gpurun_out/ab.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$*]', round(d['value']/1e9,1), round(r['avg_launch_ms'],3), round(r['epilogue_kernel_ms'],3), 'keys/line', d['config']['keys_per_128B_line'], 'GB', round(d['config']['table_bytes']/1e9,1))" || tail -3 gpurun_out/ab.err; }
run
for P in 8 16 32 128 256; do run --piece-tiles $P; done
for K in 1.5 2 2.5 3.5 4.5; do run --keys-per-bucket $K; done
for K in 2 3.5 4.5; do run --genomes 64 --genome-mb 20 --keys-per-bucket $K; done
run --genomes 64 --genome-mb 20
