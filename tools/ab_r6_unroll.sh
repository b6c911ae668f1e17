#!/bin/bash
# round 6: k_probe's skewed batch loop two batches per trip (PG_PROBE_UNROLL2) against one:  bash tools/ab_libs.sh tools/ab_r6_unroll.sh u0 u1 u0 u1
cd ${GRAFT_REPO_ROOT:-/root/repo}
PG_SHAPES=";--genomes 27 --genome-mb 40;--genomes 64 --genome-mb 20 --contigs 10 --k 31 --d 0.005;--genomes 16 --genome-mb 50;--genomes 20 --genome-mb 40;--per-genome-launches;--d 0.03" PG_STEPS=20 PG_WARMUP=5 bash tools/lines.sh $1
