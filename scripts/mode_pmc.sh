#!/bin/bash
# the two step-time modes of the 1 GiB encode: PMC counters of msd_partition_kernel per
# workspace placement, one counter set per run (the program directly behind `--`)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TRIES=${TRIES:-6}
i=0
for set in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/mode_pmc_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/mode_pmc_$i -- python scripts/dbg/mode_place.py $TRIES > gpurun_out/mode_pmc_$i.log 2>&1
  echo "== set $i: $set (rc=$?)"
  grep PLACEMENTS_MS gpurun_out/mode_pmc_$i.log
  python - <<PY
import csv,glob,collections
d="gpurun_out/mode_pmc_$i"
kt=glob.glob(d+"/*/*kernel_trace.csv"); cc=glob.glob(d+"/*/*counter_collection.csv")
if not kt or not cc:
    print("  no output", kt, cc); raise SystemExit
dur={}
for r in csv.DictReader(open(kt[0])):
    dur[r["Dispatch_Id"]]=(r["Kernel_Name"], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, int(r["Start_Timestamp"]))
cnt=collections.defaultdict(dict)
for r in csv.DictReader(open(cc[0])):
    cnt[r["Dispatch_Id"]][r["Counter_Name"]]=float(r["Counter_Value"])
rows=sorted(((v[2],k) for k,v in dur.items() if "msd_partition_kernel<false" in v[0] or "msd_partition_kernel<0" in v[0] or ("msd_partition_kernel" in v[0] and "true" not in v[0] and "<1" not in v[0])))
# 3 encodes per placement, 2 plain launches per encode -> 6 dispatches per placement (level 2, level 3 alternate)
names=sorted({c for k in cnt for c in cnt[k]})
print("  %-10s %9s %9s  %s" % ("placement","lvl2 ms","lvl3 ms","  ".join(n[:34] for n in names)))
for p in range(0, len(rows)//6):
    grp=[rows[p*6+j][1] for j in range(6)]
    l2=[dur[g][1] for g in grp[2::2]]; l3=[dur[g][1] for g in grp[3::2]]   # skip the first encode (first touch)
    vals=[]
    for nme in names:
        a=[cnt[g].get(nme,0) for g in grp[2::2]]; b=[cnt[g].get(nme,0) for g in grp[3::2]]
        vals.append("%.4g/%.4g" % (sum(a)/len(a), sum(b)/len(b)))
    print("  %-10d %9.3f %9.3f  %s" % (p, sum(l2)/len(l2), sum(l3)/len(l3), "  ".join(vals)))
PY
done
