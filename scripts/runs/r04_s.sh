#!/bin/bash
# round 4, second session: the chain rounds (tc_chain.hpp) -- forced at every dense round against the oracle, the digest
# records with the default trigger, then periodic text at 2^28 and 1 GiB with the per-step trace
mkdir -p gpurun_out
export CH="TC_SA_CHAIN=2 TC_SA_DENSE=1 TC_SA_SEG_MIN=1"
env $CH timeout -k 10 300 python tests/long/fuzz_chain.py 300 1 20000 > gpurun_out/r04s_fuzz_chain.log 2>&1; echo "fuzz_chain forced rc=$?"; tail -n 4 gpurun_out/r04s_fuzz_chain.log
env $CH TC_SA_BIN_MIN_LOG2=0 timeout -k 10 300 python tests/long/fuzz_chain.py 200 2 60000 > gpurun_out/r04s_fuzz_chain2.log 2>&1; echo "fuzz_chain forced (pairs) rc=$?"; tail -n 3 gpurun_out/r04s_fuzz_chain2.log
env $CH timeout -k 10 300 python tests/long/fuzz_long.py 150 5 60000 > gpurun_out/r04s_fuzz_long.log 2>&1; echo "fuzz_long forced rc=$?"; tail -n 3 gpurun_out/r04s_fuzz_long.log
timeout -k 10 600 python -m pytest tests/test_gpu_classes_digest.py -x -q > gpurun_out/r04s_digest.log 2>&1; echo "digest rc=$?"; tail -n 3 gpurun_out/r04s_digest.log
TC_SA_TRACE=1 timeout -k 10 300 python scripts/classes_bench.py 268435456 repeat_4KiB,repeat_1MiB > gpurun_out/r04s_periodic_28.txt 2> gpurun_out/r04s_periodic_28.err; echo "2^28 rc=$?"; cut -c1-260 gpurun_out/r04s_periodic_28.txt
timeout -k 10 300 python scripts/classes_bench.py 1073741824 repeat_4KiB,repeat_1MiB,runs_p0.9,acgtn > gpurun_out/r04s_periodic_30.txt 2> gpurun_out/r04s_periodic_30.err; echo "2^30 rc=$?"; cut -c1-260 gpurun_out/r04s_periodic_30.txt
grep -E "chain|round:" gpurun_out/r04s_periodic_28.err | head -60
