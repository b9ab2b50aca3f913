#!/bin/bash
# HBM traffic of the decode kernels (1 GiB ACGTN round trip): FETCH_SIZE / WRITE_SIZE as in pmc_traffic.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmcd_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcd_$c -- python scripts/decode_bench.py > gpurun_out/pmcd_$c.log 2>&1
done
python - <<'PY'
import csv,glob,collections
tot=collections.defaultdict(lambda: {"FETCH_SIZE":[], "WRITE_SIZE":[]})
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob("gpurun_out/pmcd_%s/*/*counter_collection.csv"%c)[0]
    for r in csv.DictReader(open(f)):
        tot[r["Kernel_Name"].split("(")[0][:50]][c].append(float(r["Counter_Value"])*1024)
print("%-52s %6s %10s %10s"%("kernel (largest dispatch)","calls","fetch GB","write GB"))
for k,d in sorted(tot.items(), key=lambda kv:-max(kv[1]["WRITE_SIZE"]+[0])-2*max(kv[1]["FETCH_SIZE"]+[0])):
    fe=2*max(d["FETCH_SIZE"]+[0]); wr=max(d["WRITE_SIZE"]+[0])
    if fe+wr < 2e8 or not any(x in k for x in ("ibwt","imtf","rle_decode","lf_")): continue
    print("%-52s %6d %10.2f %10.2f"%(k,len(d["WRITE_SIZE"]),fe/1e9,wr/1e9))
PY
