#!/bin/bash
# HBM traffic per kernel (FETCH_SIZE / WRITE_SIZE in separate passes, as the guide prescribes).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=${1:-1073741824}
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python bench.py --n $N --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$c.log 2>&1
done
python - <<'PY'
import csv,glob,collections
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob("gpurun_out/pmc_%s/*/*counter_collection.csv"%c)
    if not f: print("no file for",c); continue
    agg=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"][:50]; agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
    print("==",c,"(KB units; per dispatch avg)")
    for k,(n,v) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:12]:
        print("%-52s n=%-4d avg=%.1f MB"%(k,n,v/n/1024))
PY
