#!/bin/bash
# gpurun with a tree stamp: writes BUILD_STAMP (HEAD + a hash of the uncommitted diff + time) at the repo root before
# the snapshot is taken -- the GPU box has no .git -- so that every profile summary / bench line produced there can say
# which tree it measured (tools/gpu_profile*.sh, tools/cfg4_kernels.sh, bench.py `config.tree`).
#   bash tools/grun.sh --timeout 600 -- '<command>'
cd "$(dirname "$0")/.."
H=$(git rev-parse --short=12 HEAD 2>/dev/null || echo nogit)
if ! git diff --quiet HEAD 2>/dev/null; then H="$H+dirty:$(git diff HEAD | sha256sum | cut -c1-8)"; fi
echo "$H $(date -u +%FT%TZ)" > BUILD_STAMP
exec /usr/local/graft/bin/gpurun "$@"
