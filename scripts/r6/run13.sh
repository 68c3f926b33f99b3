#!/bin/bash
# round 6, thirteenth GPU call: labels by position in the resident kernel -- parity tests, then
# base library against new library on ONE box
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_gpu_resident.py tests/test_gpu_blocks.py tests/test_gpu_fuzz.py tests/test_gpu_default_vs_oracle.py -q -x > gpurun_out/r6/t13.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r6/t13.log
tail -3 gpurun_out/r6/t13.log
cp embiggen_amd/csrc/libgn2v.so /tmp/libgn2v_new.so
run() { name=$1; lib=$2
  cp $lib embiggen_amd/csrc/libgn2v.so
  timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r6/lab_$name.json 2> gpurun_out/r6/lab_$name.err
  python - gpurun_out/r6/lab_$name.json "$name" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(f"{sys.argv[2]:12s} value {d['value']:.4g}  kernel {r.get('kernel_pairs_per_s',0):.4g}  frac {r.get('frac',0):.3f}  launch_ms {r['avg_launch_ms']:.2f} x{r['launches']}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run base_1 scripts/r6/ab/libgn2v_base.so
run new_1 /tmp/libgn2v_new.so
run base_2 scripts/r6/ab/libgn2v_base.so
run new_2 /tmp/libgn2v_new.so
cp /tmp/libgn2v_new.so embiggen_amd/csrc/libgn2v.so
