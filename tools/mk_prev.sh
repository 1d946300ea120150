#!/bin/bash
# Builds the library of a git revision (default HEAD) next to the working-tree one, for same-box A/B runs:
#   tools/mk_prev.sh [rev]  ->  corenav_gp_amd/libcorenav_gp_prev.so   (select it with CGP_LIB=...)
set -e
rev=${1:-HEAD}
R=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/cgp_prev && git -C $R worktree prune && git -C $R worktree add -f --detach /tmp/cgp_prev $rev > /dev/null
make -C /tmp/cgp_prev/corenav_gp_amd/csrc ARCH=gfx950 > /tmp/cgp_prev_build.log 2>&1
cp /tmp/cgp_prev/corenav_gp_amd/libcorenav_gp.so $R/corenav_gp_amd/libcorenav_gp_prev.so
git -C $R worktree remove --force /tmp/cgp_prev
ls -la $R/corenav_gp_amd/libcorenav_gp_prev.so
