#!/bin/bash
# ONE line per pangenome shape from bench.py's timed region (further legs off): the parameterised form of round 2-4's
# one-off ab_*.sh / *_sweep*.sh scripts.
#
#   bash tools/lines.sh [tag]                       shapes from PG_SHAPES (';'-separated bench.py argument strings; default: configs[1])
#   PG_SHAPES="--genomes 27 --genome-mb 40;--genomes 64 --genome-mb 20" PG_EACH="--minimizer 15;--minimizer 16" bash tools/lines.sh
#                                                   every shape x every PG_EACH variant (a sweep)
#   PG_STEPS / PG_WARMUP (default 6 / 2), PG_EXTRA (arguments for every run, e.g. --per-genome-launches), PG_TIMEOUT (900 s)
#   bash tools/ab_libs.sh tools/lines.sh tagA tagB  the same over prebuilt library variants (build_variants/lib_<tag>.so)
#
# columns: [tag] [arguments] G k-mers/s | step ms | k_probe ms | statistics ms | probe ps per position | keys | table GB |
#          keys per 128-byte line | table build s | roofline frac | m (the table's minimizer length) | fraction of the keys outside their home line
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TAG=${1:-}
IFS=';' read -ra SHAPES <<< "${PG_SHAPES:-}"
[ ${#SHAPES[@]} -eq 0 ] && SHAPES=("")
IFS=';' read -ra EACH <<< "${PG_EACH:-}"
[ ${#EACH[@]} -eq 0 ] && EACH=("")
for S in "${SHAPES[@]}"; do
  for E in "${EACH[@]}"; do
    A="$S $E ${PG_EXTRA:-}"
    timeout ${PG_TIMEOUT:-900} python bench.py --steps ${PG_STEPS:-6} --warmup ${PG_WARMUP:-2} --no-cpu-baseline --no-compare --no-other-shapes \
      --no-sharded-leg --no-e2e --no-robustness --no-config5 $A 2>gpurun_out/lines.err | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; c = d['config']; n = c['positions_per_step_per_gpu']
print('[$TAG] [' + ' '.join('''$A'''.split()) + ']', round(d['value'] / 1e9, 1), '| step', round(d['ms_per_step'], 3), '| probe', round(r['avg_launch_ms'], 3),
      '| stats', round(r['epilogue_kernel_ms'], 3), '| ps/pos', round(r['avg_launch_ms'] * 1e9 / (n / c['launches_per_step']), 2), '| keys', c['table_keys'],
      '| GB', round(c['table_bytes'] / 1e9, 1), '| keys/line', c['keys_per_128B_line'], '| build s', round(c['table_build_s'], 3),
      '| frac', round(r['frac'], 3), '| m', c.get('minimizer_length'), '| spill', round(c.get('table_spill_fraction') or 0, 3))" || tail -3 gpurun_out/lines.err
  done
done
