#!/bin/bash
# round 3 profile artefacts: kernel stats of the bench (all legs), PMC traffic, the comm test
mkdir -p gpurun_out
export TC_COMMIT=$1
bash scripts/prof_brief.sh r03z > gpurun_out/r03z_summary.txt 2>&1; echo "prof rc=$?"; cat gpurun_out/r03z_summary.txt | head -40
grep '^{' gpurun_out/prof_r03z_bench.log | tail -n 1 > gpurun_out/r03z_profiled_bench.json
bash scripts/pmc_traffic.sh > gpurun_out/r03z_traffic.txt 2>&1; echo "pmc rc=$?"; cat gpurun_out/r03z_traffic.txt | tail -n 25
timeout -k 10 300 python -m pytest tests/test_gpu_comm.py tests/test_gpu_container_fused.py -x -q 2>&1 | tail -n 5
timeout -k 10 300 python bench.py --steps 20 --warmup 2 > gpurun_out/r03z_bench.json 2> gpurun_out/r03z_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03z_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["workspace_placement"], d["container"]["ms_per_step_with_container"], d["fm_count"]["ms"])
PY
