#!/bin/bash
# Build a library variant for tools/ab_libs.sh:  bash tools/build_variant.sh <tag> [source root] [-DNAME=VALUE ...]
# (source root: another checkout, e.g. a `git worktree add /tmp/base HEAD`, for the "before" of an A/B)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
SRC=$ROOT
if [ -d "$1" ]; then SRC=$1; shift; fi
mkdir -p $ROOT/build_variants
C=$SRC/panagram_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result \
  -o $ROOT/build_variants/lib_$TAG.so "$@" $C/pg_kernels.hip $C/pg_anchor.hip $C/pg_deflate.hip $C/pg_api.hip $C/pg_bgzf.cpp -lz -lpthread && echo built $TAG
