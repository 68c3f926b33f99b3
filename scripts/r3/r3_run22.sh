#!/bin/bash
# link quality of the final default on the bench graph (BA 10 M, 38 x 8, rounds of 2^23 walks in
# four groups) against the walk-ordered schedule, same walks: 2 and 4 epochs of 10^7 walks
mkdir -p gpurun_out
timeout 1500 python scripts/quality_probe.py --nodes 10000000 --walks 10000000 --epochs 2 --modes write_through,blocks:38:8 --round-walks 8388608 --group-parts 10 > gpurun_out/r3_quality22_10m_2ep.log 2>&1
tail -2 gpurun_out/r3_quality22_10m_2ep.log
timeout 1500 python scripts/quality_probe.py --nodes 10000000 --walks 10000000 --epochs 4 --modes write_through,blocks:38:8 --round-walks 8388608 --group-parts 10 > gpurun_out/r3_quality22_10m_4ep.log 2>&1
tail -2 gpurun_out/r3_quality22_10m_4ep.log
