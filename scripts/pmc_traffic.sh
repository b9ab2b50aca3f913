#!/bin/bash
# HBM traffic per kernel: FETCH_SIZE and WRITE_SIZE in separate passes (as the MI355X guide
# prescribes); FETCH_SIZE is doubled (gfx950 reports 1/2 for streaming reads: calibrated with
# scripts/pmc_calib.sh on 4/8/16-byte-per-lane streams), WRITE_SIZE is exact.  Units: KB.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=${1:-1073741824}
export TC_BENCH_PLACE=0   # (one placement: the counters are per launch, the mode does not change the bytes)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python bench.py --n $N --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$c.log 2>&1
done
# the whole step by the counters: ONE encode call and nothing else in the process (no warm-up, no extra legs)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmcstep_$c
  TC_BENCH_CONTAINER=0 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcstep_$c -- python bench.py --n $N --steps 1 --warmup 0 --no-cpu-baseline --no-fm --no-classes --no-host-path > gpurun_out/pmcstep_$c.log 2>&1
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
res={"detail": out, "commit": __import__("os").environ.get("TC_COMMIT", "?"),
     "note": "1 GiB ACGTN bench; traffic = 2*FETCH_SIZE + WRITE_SIZE of the largest dispatch of each kernel; "
             "*_bytes_per_launch = mean over that kernel's full-length launches of one step"}
# mean over the launches of one step (the partition levels differ: level 1 reads the text, not a key array)
keyonly=any("msd_partition_kernel<false, false>" in k for k in tot)   # the encode's levels move keys only (round 3)
for name,pat in (("radix_pass_kernel","radix_pass_kernel<false"),("msd_partition_kernel","msd_partition_kernel")):
    fe=[];wr=[]
    for k,d in tot.items():
        if pat in k and not (name=="msd_partition_kernel" and keyonly and ", false>" not in k):   # (not the FM leg's index builds)
            big=max(d["WRITE_SIZE"]+[0])
            for f,w in zip(sorted(d["FETCH_SIZE"])[::-1], sorted(d["WRITE_SIZE"])[::-1]):
                if w > 0.5*big: fe.append(2*f); wr.append(w)
    if wr and sum(wr) / len(wr) > 1e9:   # (full-length launches only: the refinement's small sorts reuse the kernel)
        res[name+"_bytes_per_launch"]=(sum(fe)+sum(wr))/len(wr); res[name+"_launches"]=len(wr)
# FM count (configs[3] leg of the bench): raw FETCH_SIZE of the largest launch -- 64-byte requests, so NOT doubled
for k,d in tot.items():
    if "fm_count_kernel" in k and d["FETCH_SIZE"]:
        res["fm_count_kernel_fetch_bytes_per_launch"]=max(d["FETCH_SIZE"]); res["fm_commit"]=res["commit"]
    if "rle_nib_kernel" in k and d["WRITE_SIZE"]:
        res["rle_nib_kernel_bytes_per_launch"]=2*max(d["FETCH_SIZE"]+[0])+max(d["WRITE_SIZE"])
# one encode step: every dispatch of the one-call run except the generator (and anything of torch's)
step=collections.defaultdict(float); nd=collections.Counter()
for c,mul in (("FETCH_SIZE",2.0),("WRITE_SIZE",1.0)):
    fs=glob.glob("gpurun_out/pmcstep_%s/*/*counter_collection.csv"%c)
    for r in (csv.DictReader(open(fs[0])) if fs else []):
        k=r["Kernel_Name"].split("(")[0][:60]
        if "generate" in k or "at::" in k or "elementwise" in k: continue
        step[k]+=mul*float(r["Counter_Value"])*1024
        if c=="WRITE_SIZE": nd[k]+=1
if step:
    res["step_traffic_bytes"]=sum(step.values())
    res["step_traffic_by_kernel"]={k:{"bytes":v,"dispatches":nd[k]} for k,v in sorted(step.items(), key=lambda kv:-kv[1]) if v>=5e7}
    res["step_note"]="one tc_encode_dev call of the 1 GiB record alone in the process: sum over ALL its dispatches of 2*FETCH_SIZE + WRITE_SIZE"
    print("whole step: %.2f GB in %d dispatches"%(res["step_traffic_bytes"]/1e9, sum(nd.values())))
json.dump(res, open("gpurun_out/traffic_latest.json","w"), indent=1)
PY
