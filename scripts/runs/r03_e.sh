#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_fm_pairs.py tests/test_gpu_fmindex.py tests/test_gpu_fm_multi.py tests/test_gpu_fm_config4.py tests/test_gpu_mirror.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r03e_tests.log 2>&1
echo "tests rc=$?"; tail -n 12 gpurun_out/r03e_tests.log
timeout -k 10 300 python scripts/fm_sweep.py > gpurun_out/r03e_fm_sweep.txt 2>&1; tail -n 10 gpurun_out/r03e_fm_sweep.txt
for rep in 1 2 3 4 5 6; do
  r=$(timeout -k 10 120 python scripts/dbg/mode_place.py 1 2>&1 | grep PLACEMENTS_MS); echo "default workspace, fresh process $rep: $r" | tee -a gpurun_out/r03e_modes.txt
done
timeout -k 10 400 python bench.py --steps 10 --warmup 2 > gpurun_out/r03e_bench.json 2> gpurun_out/r03e_bench.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03e_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["workspace_placement"], d["container"]["ms_per_step_with_container"], d["fm_count"]["ms"], d["fm_count"]["outside_mall"])
PY
