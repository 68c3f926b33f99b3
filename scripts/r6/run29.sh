#!/bin/bash
# round 6, call 29: the whole GPU suite twice more on the final tree (flakiness check), failures listed
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
for i in 1 2; do
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r6/flaky_$i.log 2>&1; echo "run $i rc=$?"; tail -2 gpurun_out/r6/flaky_$i.log; grep -E "^FAILED|^ERROR" gpurun_out/r6/flaky_$i.log | head
done
