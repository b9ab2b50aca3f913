#!/bin/bash
# the SMALL MTF apply kernel (container path, and the block path with TC_MTF_RLE=0) after its new chunk pass
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_container_fused.py tests/test_gpu_encode.py tests/test_gpu_comm.py tests/test_gpu_classes_digest.py -x -q > gpurun_out/r03v_a.log 2>&1; rc=$?
echo "default rc=$rc"; tail -n 3 gpurun_out/r03v_a.log
[ $rc -ne 0 ] && exit $rc
TC_MTF_RLE=0 timeout -k 10 600 python -m pytest tests/test_gpu_encode.py tests/test_gpu_msd.py tests/test_gpu_fullsize.py tests/test_gpu_soak.py -x -q > gpurun_out/r03v_b.log 2>&1; rc=$?
echo "TC_MTF_RLE=0 rc=$rc"; tail -n 3 gpurun_out/r03v_b.log
[ $rc -ne 0 ] && exit $rc
bash scripts/prof_brief.sh r03v 2>&1 | grep -E "rle_|mtf_|finish|fm_count"
grep -o '"ms_per_step[a-z_]*": [0-9.]*' gpurun_out/prof_r03v_bench.log
