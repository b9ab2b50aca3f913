#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_dec
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dec -- python scripts/dbg/dec_probe.py nrun 28 > gpurun_out/prof_dec.log 2>&1
grep "^call" gpurun_out/prof_dec.log
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_dec/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:14]:
    print("%-60s calls=%-5s avg=%8.3f ms tot=%8.2f ms"%(r["Name"][:60],r["Calls"],float(r["AverageNs"])/1e6,float(r["TotalDurationNs"])/1e6))
PY
rm -rf gpurun_out/prof_dec
