#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03l_gputests.log 2>&1
echo "gpu tests rc=$?"; tail -n 6 gpurun_out/r03l_gputests.log
bash scripts/prof_brief.sh r03l --no-fm > gpurun_out/r03l_summary.txt 2>&1; head -n 14 gpurun_out/r03l_summary.txt
