#!/bin/bash
# SQ counters of one kernel (substring $3) for one input class of classes_bench.py; $1 = class, $2 = n
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CLS=${1:-genome_like}; N=${2:-268435456}; KER=${3:-finish_kernel}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_GDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rm -rf gpurun_out/pmccls_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmccls_$i -- python scripts/classes_bench.py $N $CLS > gpurun_out/pmccls_$i.log 2>&1 || tail -3 gpurun_out/pmccls_$i.log
done
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(float); calls=collections.Counter()
for f in glob.glob("gpurun_out/pmccls_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "$KER" in r["Kernel_Name"]:
            agg[r["Counter_Name"]]+=float(r["Counter_Value"]); calls[r["Counter_Name"]]+=1
print("$CLS $KER")
for c,v in sorted(agg.items()): print("    %-24s %.4g  (%d records)"%(c,v,calls[c]))
PY
