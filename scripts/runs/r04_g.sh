#!/bin/bash
# per-round trace of the periodic class + kernel stats of two classes at 1 GiB
TC_SA_TRACE=1 timeout -k 10 300 python scripts/classes_bench.py $((1<<30)) repeat_4KiB 2>&1 | grep -v "members by\|amdgpu.ids" | sed -n '20,32p;$p' > gpurun_out/r04_g.txt
bash scripts/prof_class.sh repeat_4KiB 1073741824 > gpurun_out/r04_prof_rep.txt 2>&1
bash scripts/prof_class.sh runs_p0.9 1073741824 > gpurun_out/r04_prof_runs.txt 2>&1
