#!/bin/bash
# round 4: tied sets of <= 4096 members ordered / tabled by one workgroup -- parity, then the step
set -o pipefail
out=gpurun_out/r04_n.txt; : > $out
timeout -k 10 300 python tests/long/fuzz_long.py 400 97 120000 2>&1 | tail -n 1 | tee -a $out
TC_SA_MSD=2 TC_SA_MSD_MIN_LOG2=10 timeout -k 10 300 python tests/long/fuzz_long.py 200 98 400000 2>&1 | tail -n 1 | tee -a $out
python -m pytest tests/test_gpu_msd.py tests/test_gpu_fullsize.py tests/test_gpu_encode.py -q -x 2>&1 | tail -n 2 | tee -a $out
for i in 1 2; do python bench.py --steps 20 --warmup 2 --no-fm --no-classes --no-cpu-baseline --no-host-path 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['container']['ms_per_step_with_container'])" | tee -a $out; done
TC_SA_TINY=0 python bench.py --steps 20 --warmup 2 --no-fm --no-classes --no-cpu-baseline --no-host-path 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('TC_SA_TINY=0', d['value'], d['ms_per_step'])" | tee -a $out
