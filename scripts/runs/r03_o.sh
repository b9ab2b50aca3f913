#!/bin/bash
mkdir -p gpurun_out
export TC_COMMIT=$1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03o_gputests.log 2>&1
echo "gpu tests rc=$?"; tail -n 4 gpurun_out/r03o_gputests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 2 > gpurun_out/r03o_bench.json 2> gpurun_out/r03o_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03o_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["workspace_placement"], d["container"]["ms_per_step_with_container"], d["fm_count"]["ms"], d["stages_ms"])
PY
bash scripts/prof_brief.sh r03y > gpurun_out/r03y_summary.txt 2>&1; head -n 16 gpurun_out/r03y_summary.txt
grep '^{' gpurun_out/prof_r03y_bench.log | tail -n 1 > gpurun_out/r03y_profiled_bench.json
bash scripts/pmc_traffic.sh > gpurun_out/r03y_traffic.txt 2>&1; echo "pmc rc=$?"; tail -n 22 gpurun_out/r03y_traffic.txt
