#!/bin/bash
# rocprofv3 evidence for the GloVe SGD kernel (run through gpurun from the repo root):
#   bash scripts/profile_glove.sh <tag>
# Same recipe as profile_bench.sh: --kernel-trace --stats, then FETCH_SIZE and WRITE_SIZE in their
# own --pmc passes.  Raw output -> gpurun_out/prof_<tag>/; digest: scripts/summarize_glove.py.
set -u
TAG=${1:-r01_glove}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--nodes 1000000 --epochs 2"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$ROOTDIR/scripts/glove_probe.py" $ARGS > "$OUT/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o fetch -- python3 "$ROOTDIR/scripts/glove_probe.py" $ARGS > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o write -- python3 "$ROOTDIR/scripts/glove_probe.py" $ARGS > "$OUT/write.log" 2>&1
tail -4 "$OUT/stats.log"
du -sh "$OUT"
