#!/bin/bash
# round 5: 65..128 genomes, the inline layout (default) against the split layout (PG_WIDE_LAYOUT=split), same library, same box
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PG_SHAPES="--genomes 65 --genome-mb 10;--genomes 80 --genome-mb 10;--genomes 96 --genome-mb 10;--genomes 112 --genome-mb 10;--genomes 128 --genome-mb 10;--genomes 128 --genome-mb 10 --k 31"
for i in 1 2; do
  PG_WIDE_LAYOUT=split bash tools/lines.sh split
  PG_WIDE_LAYOUT=inline bash tools/lines.sh inline
done
