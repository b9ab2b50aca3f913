#!/bin/bash
# back-off of ineffective chain rounds: Fibonacci / Thue-Morse words with and without chain rounds, the periodic probes and classes again
mkdir -p gpurun_out
timeout -k 10 500 python scripts/dbg/fib_probe.py 28 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee gpurun_out/r04za_fib.txt
timeout -k 10 300 python scripts/dbg/chain_probe.py 2>/dev/null | tail -n 4
timeout -k 10 300 python scripts/classes_bench.py 1073741824 repeat_4KiB,repeat_1MiB,acgt_nrun,zipf_words 2>/dev/null | cut -c1-200
TC_SA_CHAIN=2 TC_SA_SEG_MIN=1 timeout -k 10 300 python tests/long/fuzz_chain.py 200 221 60000 2>&1 | tail -n 1
timeout -k 10 300 python tests/long/fuzz_chain.py 200 222 60000 2>&1 | tail -n 1
