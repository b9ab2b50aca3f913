#!/bin/bash
# calibrate FETCH_SIZE / WRITE_SIZE on known byte counts for 8- and 4-byte-per-lane streams
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/calib.py <<'PY'
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "text-compression_amd"))
import textcomp
ctx = textcomp.Context(0)
f = ctx.lib.tc_dbg_stream_bench
f.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
g = C.c_double()
for mode in (1, 2):
    for w in (16, 8, 4):
        f(ctx.handle, 4 << 30, w, mode, 1, C.byref(g))
PY
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/calib_$c -- python /tmp/calib.py > gpurun_out/calib_$c.log 2>&1
done
python - <<'PY'
import csv,glob
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob("gpurun_out/calib_%s/*/*counter_collection.csv"%c)[0]
    print("==",c)
    for r in csv.DictReader(open(f)):
        if "dbg_stream" in r["Kernel_Name"]:
            print("  %-60s %.1f MB (4096 MB moved per launch)"%(r["Kernel_Name"][:60], float(r["Counter_Value"])/1024))
PY
