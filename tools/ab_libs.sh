#!/bin/bash
# A/B of prebuilt library variants (build_variants/lib_<tag>.so) on ONE box:  bash tools/ab_libs.sh "<script> <args>" tagA tagB ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
CMD=$1; shift
cp panagram_amd/libpanagram_hip.so /tmp/lib_orig.so
for T in "$@"; do
  cp build_variants/lib_$T.so panagram_amd/libpanagram_hip.so
  echo "=== $T"
  bash $CMD $T
done
cp /tmp/lib_orig.so panagram_amd/libpanagram_hip.so
