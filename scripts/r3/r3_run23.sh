#!/bin/bash
# whole fits through the public class (gn2v_train -> C++ block path, host copy included):
# BA 1 M x 5 epochs, BA 10 M x 1 epoch; then the 10 M fit with the contextual table trained in
# node order (strided parts) instead of part-major
mkdir -p gpurun_out
timeout 1500 python scripts/fit_probe.py > gpurun_out/r3_fit23.log 2>&1
grep "epochs" gpurun_out/r3_fit23.log
GN2V_BLOCK_LAYOUT=natural timeout 1500 python scripts/fit_probe.py > gpurun_out/r3_fit23_natural.log 2>&1
grep "epochs" gpurun_out/r3_fit23_natural.log
