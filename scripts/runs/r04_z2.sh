#!/bin/bash
# chain rounds with sparse ranks: forced in every round against the oracle (sparse: no TC_SA_DENSE), the dense path again, the
# digest records, then the classes it is for
mkdir -p gpurun_out
TC_SA_CHAIN=2 TC_SA_SEG_MIN=1 timeout -k 10 300 python tests/long/fuzz_chain.py 400 181 30000 > gpurun_out/r04z2_a.log 2>&1; echo "fuzz_chain sparse forced rc=$?"; tail -n 1 gpurun_out/r04z2_a.log
TC_SA_CHAIN=2 TC_SA_SEG_MIN=1 TC_SA_ACCEL_MIN=1 timeout -k 10 300 python tests/long/fuzz_chain.py 300 182 120000 > gpurun_out/r04z2_b.log 2>&1; echo "fuzz_chain sparse forced + accel rc=$?"; tail -n 1 gpurun_out/r04z2_b.log
TC_SA_CHAIN=2 TC_SA_SEG_MIN=1 TC_SA_ACCEL_MIN=1 timeout -k 10 300 python tests/long/fuzz_long.py 300 183 120000 > gpurun_out/r04z2_c.log 2>&1; echo "fuzz_long sparse forced rc=$?"; tail -n 1 gpurun_out/r04z2_c.log
TC_SA_CHAIN=2 TC_SA_DENSE=1 TC_SA_SEG_MIN=1 timeout -k 10 300 python tests/long/fuzz_chain.py 300 184 60000 > gpurun_out/r04z2_d.log 2>&1; echo "fuzz_chain dense forced rc=$?"; tail -n 1 gpurun_out/r04z2_d.log
TC_SA_CHAIN=2 TC_SA_SEG_MIN=1 TC_SA_MSD=2 TC_SA_MSD_MIN_LOG2=10 timeout -k 10 300 python tests/long/fuzz_long.py 300 185 300000 > gpurun_out/r04z2_e.log 2>&1; echo "fuzz_long MSD + sparse chain rc=$?"; tail -n 1 gpurun_out/r04z2_e.log
timeout -k 10 600 python -m pytest tests/test_gpu_classes_digest.py -x -q > gpurun_out/r04z2_digest.log 2>&1; echo "digest rc=$?"; tail -n 2 gpurun_out/r04z2_digest.log
TC_SA_TRACE=1 timeout -k 10 300 python scripts/classes_bench.py 1073741824 acgt_nrun,genome_like,repeat_4KiB,acgtn > gpurun_out/r04z2_cls.txt 2> gpurun_out/r04z2_cls.err; cut -c1-330 gpurun_out/r04z2_cls.txt; grep "chain tables" gpurun_out/r04z2_cls.err | sort | uniq -c
