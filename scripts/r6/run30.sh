#!/bin/bash
# round 6, call 30: can two RCCL ranks share the box's one GPU?  (tests/c/fit_world_rccl.c, world = 2)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
L=$PWD/embiggen_amd/csrc
gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tests/c/fit_world_rccl.c -L $L -lgn2v -L /opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$L -Wl,-rpath,/opt/rocm/lib -o /tmp/fwr || exit 1
rm -f /tmp/job.id
(timeout 90 /tmp/fwr 0 2 /tmp/job.id > gpurun_out/r6/rccl2_rank0.log 2>&1; echo "rank0 rc=$?" >> gpurun_out/r6/rccl2_rank0.log) &
(timeout 90 /tmp/fwr 1 2 /tmp/job.id > gpurun_out/r6/rccl2_rank1.log 2>&1; echo "rank1 rc=$?" >> gpurun_out/r6/rccl2_rank1.log) &
wait
tail -5 gpurun_out/r6/rccl2_rank0.log gpurun_out/r6/rccl2_rank1.log
