#!/bin/bash
# profiles/build_ref_lib.sh <commit> <name> -- librtmi.so of another commit as profiles/ab_prev/librtmi_<name>.so (git-ignored),
# for the interleaved A/B scripts (ab_libs.py, ab_multi.py, bench_with_lib.py).  Build container only (needs git + hipcc).
set -e
C=${1:?commit}; N=${2:?name}
R=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/wt_$N
rm -rf $W; git -C $R worktree prune; git -C $R worktree add -f $W $C > /dev/null 2>&1
(cd $W && python -m raytracerfortran_amd.build > /dev/null 2>&1)
mkdir -p $R/profiles/ab_prev && cp $W/raytracerfortran_amd/librtmi.so $R/profiles/ab_prev/librtmi_$N.so
git -C $R worktree remove --force $W; ls -la $R/profiles/ab_prev/librtmi_$N.so
