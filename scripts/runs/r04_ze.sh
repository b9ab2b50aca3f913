#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_classes_digest.py -x -q 2>&1 | tail -n 3
