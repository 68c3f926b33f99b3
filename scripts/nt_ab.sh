#!/bin/bash
# same-box A/B of compile-time variants of the block kernel on the default bench:
#   bash scripts/nt_ab.sh "<flags A>" "<flags B>" [log]
A=$1; B=$2; out=${3:-gpurun_out/nt_ab.log}; : > $out
ARGS=${GN2V_AB_ARGS:-}  # extra bench arguments, e.g. GN2V_AB_ARGS="--model cbow"
run() {
    rm -f embiggen_amd/csrc/libgn2v.so
    GN2V_HIPCC_FLAGS="$1" python -c "from embiggen_amd import _lib; _lib.build()" >>$out 2>&1
    python bench.py --steps 16 --warmup 8 --no-cpu-baseline $ARGS 2>>$out | tail -1 |
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$1]', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])" | tee -a $out
}
run "$A"; run "$B"; run "$A"; run "$B"
rm -f embiggen_amd/csrc/libgn2v.so
