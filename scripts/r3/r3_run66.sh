#!/bin/bash
# margins of the norm-ratio assertion in the block-path fuzz test (3 repetitions), then the test
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys
sys.path.insert(0, "tests")
import test_gpu_fuzz as F
for rep in range(3):
    out = []
    for case in range(8):
        g, kw = F.block_path_case(case)
        p1, p0, moves, _ = F.block_path_moves(g, kw)
        out.append("%d:%s" % (case, "/".join("%.2f" % (m / s) for m, s in moves) if p0 else "-"))
    print("rep", rep, " ".join(out), flush=True)
PY
timeout 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k block_path 2>&1 | tail -3
