#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -n 8
