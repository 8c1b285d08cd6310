#!/bin/bash
# Same-box A/B of the kernels of two commits: builds <rev>'s csrc into tools/dbg/libnmarl_<tag>.so (a scratch worktree; the C-ABI must
# equal the working tree's), to be loaded through NMARL_LIB_AB next to the working tree's own library.
#   bash tools/ab_build.sh <rev> [tag=prev]       e.g.  bash tools/ab_build.sh HEAD~1
#   NMARL_LIB_AB=tools/dbg/libnmarl_prev.so python bench.py ...     vs     python bench.py ...
set -e
cd "$(dirname "$0")/.."
rev=${1:-HEAD}; tag=${2:-prev}
wt=$(mktemp -d)
git worktree add -f "$wt" "$rev" -q
( cd "$wt" && python -m deeprl_network_amd.build > /dev/null 2>&1 )
mkdir -p tools/dbg
cp "$wt/deeprl_network_amd/libnmarl_hip.so" tools/dbg/libnmarl_$tag.so
git worktree remove --force "$wt"
echo tools/dbg/libnmarl_$tag.so
