#!/bin/bash
# round 4: the evidence run on one box -- full -m gpu suite, the bench line as the driver runs it, kernel stats of the
# profiled bench, HBM traffic per kernel and of the whole step, the 12 input classes at 1 GiB.  $1 = commit
mkdir -p gpurun_out
export TC_COMMIT=$1
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04o_gputests.log 2>&1
echo "gpu tests rc=$?"; tail -n 3 gpurun_out/r04o_gputests.log
timeout -k 10 600 python bench.py --steps 20 --warmup 2 > gpurun_out/r04_z_bench1g.json 2> gpurun_out/r04o_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_z_bench1g.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["container"]["ms_per_step_with_container"], d["fm_count"]["ms"])
print({k: (v.get("encode_ms"), v.get("round_trip_exact")) for k, v in d["classes"].items() if isinstance(v, dict)})
print(d["host_path"])
PY
bash scripts/prof_brief.sh r04z --no-fm --no-classes --no-host-path > gpurun_out/r04_z_bench1g_summary.txt 2>&1; head -n 16 gpurun_out/r04_z_bench1g_summary.txt
grep '^{' gpurun_out/prof_r04z_bench.log | tail -n 1 > gpurun_out/r04_z_bench1g_profiled.json
cp gpurun_out/prof_r04z/*/*kernel_stats.csv gpurun_out/r04_z_bench1g_kernel_stats.csv
bash scripts/pmc_traffic.sh > gpurun_out/r04_z_traffic.txt 2>&1; echo "pmc rc=$?"; tail -n 24 gpurun_out/r04_z_traffic.txt
timeout -k 10 800 python scripts/classes_bench.py 1073741824 > gpurun_out/r04_input_classes_1g.txt 2>/dev/null; cut -c1-150 gpurun_out/r04_input_classes_1g.txt
