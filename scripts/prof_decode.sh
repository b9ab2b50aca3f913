#!/bin/bash
# rocprofv3 kernel stats of the 1 GiB round trip (scripts/decode_bench.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_dec
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dec -- python scripts/decode_bench.py ${1:-1073741824} > gpurun_out/prof_dec.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_dec/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:24]:
    n=r["Name"]
    if any(k in n for k in ("ibwt","imtf","rle_","scan64","radix_pass","sym_hist")):
        print("%-58s calls=%-4s avg=%8.3f ms tot=%8.2f"%(n[:58],r["Calls"],float(r["AverageNs"])/1e6,float(r["TotalDurationNs"])/1e6))
PY
tail -1 gpurun_out/prof_dec.log
