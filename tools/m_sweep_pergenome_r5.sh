#!/bin/bash
# round 5: minimizer length by shape with ONE LAUNCH PER GENOME (no co-scheduling partner: what a single-anchor `panagram index`, the run_anchor CLI
# with one FASTA or py_kmc_api-style callers get) — what minimizer_length's second cost table (wcost_one, pg_device.h) is fitted on
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PG_EXTRA="--per-genome-launches" PG_STEPS=4
PG_SHAPES="--genomes 8 --genome-mb 100;--genomes 8 --genome-mb 30;--genomes 27 --genome-mb 40;--genomes 8 --genome-mb 300" PG_EACH="--minimizer 14;--minimizer 15;--minimizer 16;--minimizer 17;--minimizer 18" bash tools/lines.sh k21
PG_SHAPES="--genomes 8 --genome-mb 100 --k 25" PG_EACH="--minimizer 18;--minimizer 19;--minimizer 20;--minimizer 21" bash tools/lines.sh k25
echo "## the library's own choice (cosched = 1 is set by bench.py for this mode)"
PG_SHAPES="--genomes 8 --genome-mb 100;--genomes 8 --genome-mb 30;--genomes 27 --genome-mb 40;--genomes 8 --genome-mb 300;--genomes 8 --genome-mb 100 --k 25" bash tools/lines.sh lib
