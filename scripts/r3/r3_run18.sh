#!/bin/bash
# link quality with records of single-pair runs (BA 1 M in 32 x 8 cells, rounds of 2^18 walks: 1.3 pairs
# per cell and centre): pair-per-group path vs run-major loop, same walks
mkdir -p gpurun_out
for tag in ppg noppg; do
  flags=""; [ $tag = noppg ] && flags="-DGN2V_BLOCK_NO_PPG"
  GN2V_HIPCC_FLAGS="$flags" python -c "from embiggen_amd import _lib; _lib.build(force=True)" || exit 1
  timeout 900 python scripts/quality_probe.py --nodes 1000000 --epochs 3 --modes blocks:32:8 --round-walks 262144 > gpurun_out/r3_quality18_$tag.log 2>&1
  tail -1 gpurun_out/r3_quality18_$tag.log
done
python -c "from embiggen_amd import _lib; _lib.build(force=True)"
bash scripts/profile_bench.sh r03 > gpurun_out/r3_prof_r03.log 2>&1
tail -2 gpurun_out/r3_prof_r03.log
bash scripts/profile_bench.sh r03_100m --nodes 100000000 --steps 8 --warmup 8 > gpurun_out/r3_prof_r03_100m.log 2>&1
tail -2 gpurun_out/r3_prof_r03_100m.log
