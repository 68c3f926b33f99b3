#!/bin/bash
# what the driver runs at round end: GPU suite, smoke, the default bench line
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r3_gputests65.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_gputests65.log
tail -4 gpurun_out/r3_gputests65.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r3_bench65_default.json 2> gpurun_out/r3_bench65_default.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r3_bench65_default.json") if l.startswith("{")][-1]); r=d["roofline"]
print("value %.4e ms/step %.1f frac_hbm %s frac %.3f sched %.3f mem %.0f cpu %.3e"%(d["value"], d["ms_per_step"], r["frac_hbm"], r["frac"], r["frac_scheduled"], d["hbm_peak_gb"]["torch_allocated"], d["cpu_baseline"]["value"]))
PY
timeout 900 python bench.py --model cbow --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); r=l['roofline']; print('cbow %.4e frac %.3f frac_hbm %s %s'%(l['value'], r['frac'], r['frac_hbm'], r['kernel']))"
