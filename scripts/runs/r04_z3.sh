#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 200 python scripts/dbg/dec_probe.py nrun 2>&1 | grep -v amdgpu.ids
TC_IBWT_LF=0 timeout -k 10 200 python scripts/dbg/dec_probe.py nrun 2>&1 | grep -v amdgpu.ids | tail -2
