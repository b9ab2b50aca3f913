#!/bin/bash
# BASELINE configs[3] evidence: fm_bench.py plain, under timeout -k 10 240 rocprofv3 --kernel-trace --stats, and with
# FETCH_SIZE / WRITE_SIZE in separate --pmc passes (units KB; FETCH_SIZE x2 on gfx950, see
# profiles/r01_pmc_calibration.txt).  Writes gpurun_out/r02_fm_count.txt.  $1 = text bytes, $2 = patterns
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=${1:-268435456}; NP=${2:-10000000}
O=gpurun_out/r02_fm_count.txt
timeout -k 10 240 python3 scripts/fm_bench.py $N $NP 5 > gpurun_out/fm_bench.log 2>&1 || { tail -20 gpurun_out/fm_bench.log; exit 1; }
rm -rf gpurun_out/fm_prof gpurun_out/fm_pmc_FETCH_SIZE gpurun_out/fm_pmc_WRITE_SIZE
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fm_prof -- python3 scripts/fm_bench.py $N $NP 3 > gpurun_out/fm_prof.log 2>&1 || { tail -20 gpurun_out/fm_prof.log; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/fm_pmc_$c -- python3 scripts/fm_bench.py $N $NP 1 > gpurun_out/fm_pmc_$c.log 2>&1 || { tail -20 gpurun_out/fm_pmc_$c.log; exit 1; }
done
{
echo "# FM-index count, BASELINE configs[3]: $NP x 100-byte ACGTN patterns, text of $N bytes, one MI355X"
echo "## scripts/fm_bench.py (no profiler)"
cat gpurun_out/fm_bench.log
echo
echo "## timeout -k 10 240 rocprofv3 --kernel-trace --stats -- python3 scripts/fm_bench.py $N $NP 3   (FM kernels)"
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/fm_prof/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("fm_","radix","finish","hist")) and float(r["TotalDurationNs"])>2e5:
        print("%-60s calls=%-4s avg=%9.3f ms min=%9.3f max=%9.3f"%(r["Name"][:60],r["Calls"],float(r["AverageNs"])/1e6,float(r["MinNs"])/1e6,float(r["MaxNs"])/1e6))
PY
echo
echo "## HBM traffic of fm_count_kernel (largest dispatch = the $NP-pattern batch): 2 x FETCH_SIZE + WRITE_SIZE"
python3 - <<'PY'
import csv,glob
v={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob("gpurun_out/fm_pmc_%s/*/*counter_collection.csv"%c)[0]
    v[c]=max([float(r["Counter_Value"])*1024 for r in csv.DictReader(open(f)) if "fm_count_kernel" in r["Kernel_Name"]]+[0])
fe,wr=2*v["FETCH_SIZE"],v["WRITE_SIZE"]
print("fetch %.2f GB  write %.2f GB  traffic %.2f GB per launch"%(fe/1e9,wr/1e9,(fe+wr)/1e9))
PY
} > $O
cat $O
