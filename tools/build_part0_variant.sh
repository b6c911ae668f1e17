#!/bin/bash
# A library variant that differs in pg_anchor.hip's part 0 only (the statistics kernels, the launchers, the w = 3 / 4 probes: defines on the
# command line), linked with the objects of the last `python panagram_amd/build.py`:  bash tools/build_part0_variant.sh <tag> [-DNAME=VALUE ...]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/panagram_amd/build
mkdir -p $ROOT/build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result \
  -c -o /tmp/pg_anchor_p0_$TAG.o -DPG_ANCHOR_PART=0 "$@" $ROOT/panagram_amd/csrc/pg_anchor.hip || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build_variants/lib_$TAG.so $O/pg_anchor_p2.o $O/pg_anchor_p1.o /tmp/pg_anchor_p0_$TAG.o \
  $O/pg_api.o $O/pg_kernels.o $O/pg_deflate.o $O/pg_bgzf.o -lz -lpthread && echo built $TAG
