#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_classes_digest.py -x -q -k "periodic" > gpurun_out/r04z4_digest.log 2>&1; echo "digest rc=$?"; tail -n 3 gpurun_out/r04z4_digest.log
