#!/bin/bash
# the chain scan's phase B with eight loads in flight, blockany over huge row blocks: parity (forced chain rounds, dense and
# sparse, small and large h), the digest records, the classes
mkdir -p gpurun_out
TC_SA_CHAIN=2 TC_SA_DENSE=1 TC_SA_SEG_MIN=1 timeout -k 10 300 python tests/long/fuzz_chain.py 400 211 60000 > gpurun_out/r04z9_a.log 2>&1; echo "dense rc=$?"; tail -n 1 gpurun_out/r04z9_a.log
TC_SA_CHAIN=2 TC_SA_SEG_MIN=1 TC_SA_ACCEL_MIN=1 timeout -k 10 300 python tests/long/fuzz_chain.py 400 212 200000 > gpurun_out/r04z9_b.log 2>&1; echo "sparse rc=$?"; tail -n 1 gpurun_out/r04z9_b.log
TC_SA_CHAIN=2 TC_SA_DENSE=1 TC_SA_SEG_MIN=1 TC_SA_H_START=4 timeout -k 10 300 python tests/long/fuzz_chain.py 200 213 400000 > gpurun_out/r04z9_c.log 2>&1; echo "dense, h from 4 (many row blocks) rc=$?"; tail -n 1 gpurun_out/r04z9_c.log
timeout -k 10 600 python -m pytest tests/test_gpu_classes_digest.py -x -q > gpurun_out/r04z9_digest.log 2>&1; echo "digest rc=$?"; tail -n 2 gpurun_out/r04z9_digest.log
TC_SA_TRACE=1 timeout -k 10 300 python scripts/classes_bench.py 1073741824 repeat_4KiB 2> gpurun_out/r04z9_trace.err | cut -c1-200; grep "chain round: codes" gpurun_out/r04z9_trace.err | tail -n 2
timeout -k 10 300 python scripts/classes_bench.py 1073741824 repeat_4KiB,repeat_1MiB,acgt_nrun 2>/dev/null | cut -c1-200
