#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03g_gputests.log 2>&1
echo "gpu tests rc=$?"; tail -n 6 gpurun_out/r03g_gputests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 2 > gpurun_out/r03g_bench.json 2> gpurun_out/r03g_bench.err; echo "bench rc=$?"
tail -n 3 gpurun_out/r03g_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03g_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["workspace_placement"], d["container"]["ms_per_step_with_container"], d["fm_count"]["ms"])
PY
