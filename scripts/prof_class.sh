#!/bin/bash
# rocprofv3 kernel stats of one input class of classes_bench.py; $1 = class, $2 = n
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CLS=${1:-zipf_words}; N=${2:-268435456}
rm -rf gpurun_out/prof_cls_$CLS
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cls_$CLS -- python scripts/classes_bench.py $N $CLS > gpurun_out/prof_cls_${CLS}.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_cls_$CLS/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print("%-60s calls=%-5s avg=%8.3f ms tot=%8.2f ms %4.1f%%"%(r["Name"][:60],r["Calls"],float(r["AverageNs"])/1e6,float(r["TotalDurationNs"])/1e6,100*float(r["TotalDurationNs"])/tot))
PY
tail -2 gpurun_out/prof_cls_${CLS}.log
