for mb in 2 5 10 25 100; do
python bench.py --genomes 8 --genome-mb $mb --steps 20 --warmup 3 --no-compare --no-cpu-baseline --no-other-shapes --no-sharded-leg --no-e2e --no-robustness 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print($mb, d['value']/1e9, d['ms_per_step'], d['roofline'].get('kernel_ms'), d['config'].get('table_bytes'))
"
done
