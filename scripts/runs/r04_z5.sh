#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_generators.py tests/test_gpu_container_fused.py tests/test_gpu_decode.py -x -q > gpurun_out/r04z5_tests.log 2>&1; echo "tests rc=$?"; tail -n 2 gpurun_out/r04z5_tests.log
timeout -k 10 300 python tests/long/fuzz_raw.py 400 191 > gpurun_out/r04z5_raw.log 2>&1; echo "fuzz_raw rc=$?"; tail -n 1 gpurun_out/r04z5_raw.log
timeout -k 10 300 python tests/long/fuzz_long.py 300 192 120000 > gpurun_out/r04z5_long.log 2>&1; echo "fuzz_long rc=$?"; tail -n 1 gpurun_out/r04z5_long.log
timeout -k 10 400 python bench.py --steps 5 --warmup 1 --no-fm --no-host-path --no-cpu-baseline > gpurun_out/r04z5_bench.json 2> gpurun_out/r04z5_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04z5_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
for k,v in d["classes"].items():
    if isinstance(v,dict): print(k, v)
PY
timeout -k 10 300 python scripts/classes_bench.py 1073741824 acgt_nrun 2>/dev/null | cut -c1-330
