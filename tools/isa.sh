#!/bin/bash
# gfx950 assembly of one kernel source (device side only):  bash tools/isa.sh <out.s> [source.hip] [-DNAME=VALUE ...]
# then  python tools/isa_count.py <out.s> '<kernel name prefix>'  counts the instructions of a kernel by kind
OUT=$1; shift
SRC=${GRAFT_REPO_ROOT:-/root/repo}/panagram_amd/csrc/pg_anchor.hip
if [ -f "$1" ]; then SRC=$1; shift; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -o $OUT "$@" $SRC
