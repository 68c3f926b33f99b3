#!/bin/bash
# rocprofv3 evidence for the walk sampler (run through gpurun from the repo root):
#   bash scripts/profile_walks.sh <tag>
# --kernel-trace --stats, then FETCH_SIZE in its own --pmc pass, of scripts/typed_walk_probe.py.
set -u
TAG=${1:-r01_walks}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$ROOTDIR/scripts/typed_walk_probe.py" > "$OUT/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o fetch -- python3 "$ROOTDIR/scripts/typed_walk_probe.py" > "$OUT/fetch.log" 2>&1
tail -12 "$OUT/stats.log"
