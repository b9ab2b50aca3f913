#!/bin/bash
# round 4: the key round (whole buckets of the MSD way's big finish ordered by the key's remaining 32 bits) -- forced at small sizes against the oracle, digests, then genome-like at 1 GiB
set -o pipefail
out=gpurun_out/r04_r.txt; : > $out
TC_SA_MSD=2 TC_SA_MSD_MIN_LOG2=10 TC_SA_MSD_BIG=1 TC_SA_SEG_MIN=1 TC_SA_ACCEL_MIN=1 timeout -k 10 400 python tests/long/fuzz_long.py 400 121 300000 2>&1 | tail -n 1 | tee -a $out
TC_SA_MSD=2 TC_SA_MSD_MIN_LOG2=10 TC_SA_MSD_BIG=1 TC_SA_SEG_MIN=1 timeout -k 10 400 python tests/long/fuzz_long.py 300 122 400000 2>&1 | tail -n 1 | tee -a $out
TC_SA_MSD=2 TC_SA_MSD_MIN_LOG2=10 TC_SA_SEG_MIN=1 TC_SA_ACCEL_MIN=1 timeout -k 10 400 python tests/long/fuzz_long.py 200 123 300000 2>&1 | tail -n 1 | tee -a $out
python -m pytest tests/test_gpu_classes_digest.py tests/test_gpu_msd.py tests/test_gpu_fmindex.py -q -x 2>&1 | tail -n 3 | tee -a $out
TC_SA_TRACE=1 timeout -k 10 300 python scripts/classes_bench.py 1073741824 genome_like 2>&1 | grep -v "members by\|amdgpu" | tail -n 24 | cut -c1-160 | tee -a $out
timeout -k 10 300 python scripts/classes_bench.py 1073741824 genome_like,acgt4 2>&1 | grep -v amdgpu | cut -c1-200 | tee -a $out
