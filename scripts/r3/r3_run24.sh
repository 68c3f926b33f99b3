#!/bin/bash
# instruction mix and wait cycles of the CBOW kernel (and, for comparison, the block kernel)
mkdir -p gpurun_out/r3_diag3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for ctr in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/r3_diag3/cbow_$i -o c -- python3 $R/bench.py --model cbow --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r3_diag3/cbow_$i.log 2>&1
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/r3_diag3/sg_$i -o c -- python3 $R/bench.py --steps 2 --warmup 0 --round-walks 2097152 --no-cpu-baseline > $R/gpurun_out/r3_diag3/sg_$i.log 2>&1
done
cd $R
python - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob("gpurun_out/r3_diag3/*/*counter_collection.csv")):
    agg=collections.defaultdict(float); n=collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if "cbow_lazy_kernel" in r["Kernel_Name"] or "sgns_block_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    print(f.split("/")[-2], {k:"%.4e"%v for k,v in agg.items()}, max(n.values()) if n else 0)
PY
