#!/bin/bash
# A library variant that differs in pg_deflate.hip only (defines on the command line), linked with the objects of the last
# `python panagram_amd/build.py` (seconds instead of minutes):  bash tools/build_deflate_variant.sh <tag> [-DNAME=VALUE ...]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/panagram_amd/build
mkdir -p $ROOT/build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result \
  -c -o /tmp/pg_deflate_$TAG.o "$@" $ROOT/panagram_amd/csrc/pg_deflate.hip || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build_variants/lib_$TAG.so $O/pg_anchor_p2.o $O/pg_anchor_p1.o $O/pg_anchor_p0.o \
  $O/pg_api.o $O/pg_kernels.o /tmp/pg_deflate_$TAG.o $O/pg_bgzf.o -lz -lpthread && echo built $TAG
