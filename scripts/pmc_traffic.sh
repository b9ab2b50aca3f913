#!/bin/bash
# HBM traffic per kernel: FETCH_SIZE and WRITE_SIZE in separate passes (as the MI355X guide
# prescribes); FETCH_SIZE is doubled (gfx950 reports 1/2 for streaming reads: calibrated with
# scripts/pmc_calib.sh on 4/8/16-byte-per-lane streams), WRITE_SIZE is exact.  Units: KB.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=${1:-1073741824}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python bench.py --n $N --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$c.log 2>&1
done
python - <<'PY'
import csv,glob,collections,json
N=1073741824
tot=collections.defaultdict(lambda: {"FETCH_SIZE":[], "WRITE_SIZE":[]})
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob("gpurun_out/pmc_%s/*/*counter_collection.csv"%c)[0]
    for r in csv.DictReader(open(f)):
        tot[r["Kernel_Name"].split("(")[0][:60]][c].append(float(r["Counter_Value"])*1024)
out={}
print("%-62s %8s %12s %12s %12s"%("kernel (largest dispatch)","calls","fetch GB","write GB","traffic GB"))
for k,d in sorted(tot.items(), key=lambda kv:-max(kv[1]["WRITE_SIZE"]+[0])-2*max(kv[1]["FETCH_SIZE"]+[0])):
    fe=2*max(d["FETCH_SIZE"]+[0]); wr=max(d["WRITE_SIZE"]+[0])
    if fe+wr < 1e8: continue
    print("%-62s %8d %12.2f %12.2f %12.2f"%(k,len(d["WRITE_SIZE"]),fe/1e9,wr/1e9,(fe+wr)/1e9))
    out[k]={"fetch_bytes":fe,"write_bytes":wr}
rp=[v for k,v in out.items() if "radix_pass_kernel<false" in k]
if rp:
    json.dump({"radix_pass_kernel_bytes_per_launch": rp[0]["fetch_bytes"]+rp[0]["write_bytes"], "detail": out,
               "note": "largest dispatch of each kernel, 1 GiB ACGTN bench; traffic = 2*FETCH_SIZE + WRITE_SIZE"},
              open("gpurun_out/traffic_latest.json","w"), indent=1)
PY
