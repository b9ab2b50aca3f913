#!/bin/bash
# round 4: level 1 with its key-generation image made one tile ahead -- parity (MSD tests, 1 GiB digest, forced-MSD fuzz), then the step
set -o pipefail
out=gpurun_out/r04_m.txt; : > $out
python -m pytest tests/test_gpu_msd.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -n 3 | tee -a $out
TC_SA_MSD=2 TC_SA_MSD_MIN_LOG2=10 timeout -k 10 300 python tests/long/fuzz_long.py 200 95 400000 2>&1 | tail -n 1 | tee -a $out
TC_SA_MSD=2 TC_SA_MSD_MIN_LOG2=10 TC_SA_MSD_KEYONLY=2 timeout -k 10 300 python tests/long/fuzz_long.py 200 96 400000 2>&1 | tail -n 1 | tee -a $out
python bench.py --steps 20 --warmup 2 --no-fm --no-classes --no-cpu-baseline --no-host-path > gpurun_out/r04_m_bench.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r04_m_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['stages_ms'])" | tee -a $out
bash scripts/prof_brief.sh r04m --no-fm --no-classes --no-host-path 2>&1 | head -14 | tee -a $out
