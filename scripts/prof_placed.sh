#!/bin/bash
# rocprofv3 kernel-trace of the bench WITH its placement search; per-kernel averages over the launches of the
# warm-up + timed steps only (the encodes of the search sit in front of them in the trace and are cut off); $1 = tag
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-placed}; STEPS=${2:-3}
rm -rf gpurun_out/prof_$TAG
TC_BENCH_PLACE=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_$TAG -- python bench.py --steps $STEPS --warmup 1 --no-cpu-baseline > gpurun_out/prof_${TAG}_bench.log 2>&1
python - <<PY
import csv,glob,json,collections
f=glob.glob("gpurun_out/prof_$TAG/*/*kernel_trace.csv")[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
d=json.loads([l for l in open("gpurun_out/prof_${TAG}_bench.log") if l.startswith("{")][-1])
r=d["roofline"]
print("# bench line of this profiled run: %.0f MB/s, %.3f ms/step, %s avg launch %.4f ms (HIP events) = frac %.4f; workspace_placement %s" % (d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"], r["frac"], d["workspace_placement"]))
enc=$STEPS+1
# the last enc encodes: everything from the enc-th last hist256_kernel launch on
starts=[i for i,x in enumerate(rows) if x["Kernel_Name"].startswith("hist256_kernel")]
tail=rows[starts[-enc]:]
agg=collections.OrderedDict()
for x in tail:
    k=x["Kernel_Name"].split("(")[0][:58]
    agg.setdefault(k,[]).append((int(x["End_Timestamp"])-int(x["Start_Timestamp"]))/1e6)
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:14]:
    print("%-58s calls=%-4d avg=%8.3f ms min=%8.3f max=%8.3f tot=%8.2f"%(k,len(v),sum(v)/len(v),min(v),max(v),sum(v)))
p=[x for k,v in agg.items() if "msd_partition" in k for x in v]
print("# msd_partition_kernel, mean over the %d launches of these %d encodes: %.4f ms" % (len(p), enc, sum(p)/len(p)))
PY
