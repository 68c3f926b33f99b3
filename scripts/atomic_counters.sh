#!/bin/bash
# (round 6) The L2 atomic ceiling as evidence: rocprofv3 --pmc passes of the L2 (TCC) counters on
#   (a) scripts/atomic_probe.hip built as a program (what the chip retires: 3.1-3.3e11 f32 adds/s)
#   (b) bench.py's resident kernel (the kernel priced against that ceiling),
# two counters per pass, --kernel-trace for names and durations, no other trace domain, the
# program itself after `--`.  Digest -> gpurun_out/atomic_counters.json (per launch: counter sums,
# duration, atomics per TCC channel-cycle, TCC busy share, tag-stall share).
#   bash scripts/atomic_counters.sh [bench args...]
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/atomic_counters
mkdir -p "$OUT"
hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics "$ROOTDIR/scripts/atomic_probe.hip" -o "$OUT/atomic_probe" || exit 1
cd /tmp; export TMPDIR=/tmp
ARGS="--steps 8 --warmup 0 --no-cpu-baseline $*"
i=0
for grp in "TCC_ATOMIC_sum TCC_ATOMIC_SECTORS_sum" "TCC_BUSY_sum TCC_CYCLE_sum" "TCC_TAG_STALL_sum TCC_REQ_sum" "TCC_EA0_ATOMIC_sum TCC_EA0_ATOMIC_LEVEL_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" "TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCC_WRITE_sum"; do
  for rows in 100000 10000000; do
    timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/probe_${rows}_g$i" -o c -- "$OUT/atomic_probe" $rows 2000 > "$OUT/probe_${rows}_g$i.log" 2>&1
  done
  timeout 900 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/bench_g$i" -o c -- python3 "$ROOTDIR/bench.py" $ARGS > "$OUT/bench_g$i.log" 2>&1
  i=$((i+1))
done
python3 "$ROOTDIR/scripts/summarize_atomic_counters.py" "$OUT" > "$ROOTDIR/gpurun_out/atomic_counters.json"
cat "$ROOTDIR/gpurun_out/atomic_counters.json"
