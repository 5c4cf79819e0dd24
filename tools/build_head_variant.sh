#!/bin/bash
# Builds the COMMITTED tree (HEAD) as a library variant for an A/B against the working tree:
#   tools/build_head_variant.sh prev   ->  verbatim-rag_amd/libvrag_amd_prev.so (+ the harness library), then tools/ab_libs.sh 2 "" prev
TAG=${1:-prev}
W=/tmp/vrag_head_$TAG
rm -rf "$W"; git worktree prune; git worktree add -f --detach "$W" HEAD > /dev/null 2>&1 || exit 1
( cd "$W" && VRAG_BUILD_VARIANT=$TAG python verbatim-rag_amd/build.py > /tmp/build_$TAG.log 2>&1 ) || { tail -5 /tmp/build_$TAG.log; exit 1; }
cp "$W/verbatim-rag_amd/libvrag_amd_$TAG.so" "$W/verbatim-rag_amd/libvrag_amd_dbg_$TAG.so" verbatim-rag_amd/
git worktree remove --force "$W"
ls -la verbatim-rag_amd/libvrag_amd_$TAG.so
