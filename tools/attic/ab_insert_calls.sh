#!/bin/bash
# per-call wall times of the default build, for tools/ab_libs.sh:  bash tools/ab_libs.sh tools/ab_insert_calls.sh tagA tagB
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python tools/insert_calls.py $1 2>&1 | grep "^\["
