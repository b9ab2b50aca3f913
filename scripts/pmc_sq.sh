#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=${1:-268435456}
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT" "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY" "GRBM_TA_BUSY GRBM_TC_BUSY" "GRBM_EA_BUSY GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmcsq_$i -- python bench.py --n $N --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmcsq_$i.log 2>&1 || tail -3 gpurun_out/pmcsq_$i.log
done
python - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob("gpurun_out/pmcsq_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:44]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,d in agg.items():
    if any(x in k for x in ("radix_pass_kernel<false","finish_kernel","rle_encode","mtf_nib_apply","msd_")):
        print(k); 
        for c,v in sorted(d.items()): print("    %-24s %.4g"%(c,v))
PY
