#!/bin/bash
# round 4: why does FM count fall from 75-80 to 49 G steps/s between 2^29 and 2^30 bytes of text?  Address translation and
# cache counters of fm_count_kernel at both sizes (one counter set per run; rocprofv3 --pmc serialises kernels, the rates
# of these runs are not measurements)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_fm_pmc.txt; : > $out
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST" "TCP_TCC_READ_REQ TCC_HIT TCC_MISS TCC_EA0_RDREQ" "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_TAG_STALL TCC_EA0_RDREQ_LEVEL"; do
  i=$((i+1))
  for lg in 29 30; do
    rm -rf gpurun_out/fmpmc_${i}_$lg
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/fmpmc_${i}_$lg -- python scripts/fm_sweep.py 4000000 $lg > gpurun_out/fmpmc_${i}_$lg.log 2>&1 || tail -2 gpurun_out/fmpmc_${i}_$lg.log >> $out
  done
done
python - >> $out <<'PY'
import csv,glob,collections
for lg in (29,30):
    agg=collections.defaultdict(list)
    for f in glob.glob("gpurun_out/fmpmc_*_%d/*/*counter_collection.csv"%lg):
        for r in csv.DictReader(open(f)):
            if "fm_count_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("text 2^%d: fm_count_kernel, per launch (mean of %d launches)"%(lg, max(len(v) for v in agg.values()) if agg else 0))
    for c,v in sorted(agg.items()): print("    %-36s %.4g"%(c,sum(v)/len(v)))
PY
