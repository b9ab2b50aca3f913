#!/bin/bash
# A/B the finish kernel time of prebuilt libtextcomp_<tag>.so variants under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp text-compression_amd/libtextcomp.so /tmp/libtextcomp_orig.so
for tag in "$@" default; do
  if [ "$tag" = default ]; then cp /tmp/libtextcomp_orig.so text-compression_amd/libtextcomp.so; else cp text-compression_amd/libtextcomp_$tag.so text-compression_amd/libtextcomp.so; fi
  rm -rf gpurun_out/prof_ab_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ab_$tag -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_ab_$tag/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "finish" in r["Name"]: print("$tag", r["Name"][:40], "avg %.3f ms"%(float(r["AverageNs"])/1e6))
PY
done
