#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L > gpurun_out/r03_counters.txt 2>&1
echo "counters rc=$? lines=$(wc -l < gpurun_out/r03_counters.txt)"
timeout -k 10 300 python scripts/fm_sweep.py > gpurun_out/r03b_fm_sweep.txt 2>&1
echo "sweep rc=$?"; cat gpurun_out/r03b_fm_sweep.txt | tail -n 12
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03b_gputests.log 2>&1
echo "gpu tests rc=$?"; tail -n 12 gpurun_out/r03b_gputests.log
