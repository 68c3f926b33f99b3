#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   bash scripts/profile_bench.sh <tag> [bench args...]
# 1) --kernel-trace --stats (per-kernel time), 2) --pmc FETCH_SIZE, 3) --pmc WRITE_SIZE in their
# own passes (never combined with other trace domains).  Raw output -> gpurun_out/prof_<tag>/.
set -u
TAG=${1:-r01}; shift || true
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# the driver's own command line (bench.py's defaults: --steps 20 --warmup 5) unless the caller
# names other step counts; the summary works on totals over all launches, whatever their sizes
case " $* " in
  *" --steps "*) ARGS="--no-cpu-baseline $*" ;;
  *) ARGS="--steps 20 --warmup 5 --no-cpu-baseline $*" ;;
esac
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$ROOTDIR/bench.py" $ARGS > "$OUT/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o fetch -- python3 "$ROOTDIR/bench.py" $ARGS > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o write -- python3 "$ROOTDIR/bench.py" $ARGS > "$OUT/write.log" 2>&1
find "$OUT" -type f | head -50
du -sh "$OUT"
# calibration passes: touch_rows_kernel moves a known number of bytes with the same access shape
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/cal_fetch" -o cal_fetch -- python3 "$ROOTDIR/bench.py" --calibrate $ARGS > "$OUT/cal_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/cal_write" -o cal_write -- python3 "$ROOTDIR/bench.py" --calibrate $ARGS > "$OUT/cal_write.log" 2>&1
grep -h calibration "$OUT/cal_fetch.log" "$OUT/cal_write.log"
