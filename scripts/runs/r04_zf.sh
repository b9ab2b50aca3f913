#!/bin/bash
# what the driver runs at round end, on the round's last commit: smoke(), the full -m gpu suite, the default bench
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04zf_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -n 2 gpurun_out/r04zf_gputests.log
timeout -k 10 600 python bench.py > gpurun_out/r04zf_bench.json 2> gpurun_out/r04zf_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04zf_bench.json").read().strip().splitlines()[-1])
print(d["metric"], d["value"], d["unit"], d["ms_per_step"], d["n_gpus"], d["steps"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
print({k: (v.get("encode_ms"), v.get("chain_rounds"), v.get("round_trip_exact")) for k, v in d["classes"].items() if isinstance(v, dict)})
PY
