#!/bin/bash
# rocprofv3 kernel-trace of the bench; prints per-kernel avg times; $1 = tag
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-x}; shift
rm -rf gpurun_out/prof_$TAG
export TC_BENCH_PLACE=${TC_BENCH_PLACE:-0}   # (the placement search would mix the encodes of rejected placements into the averages)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/prof_${TAG}_bench.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_$TAG/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-58s calls=%-4s avg=%8.3f ms tot=%8.2f"%(r["Name"][:58],r["Calls"],float(r["AverageNs"])/1e6,float(r["TotalDurationNs"])/1e6))
PY
