#!/bin/bash
# round 4, second session: chain rounds with the middle member as the reference; the bounded far scan of the MTF list
mkdir -p gpurun_out
export CH="TC_SA_CHAIN=2 TC_SA_DENSE=1 TC_SA_SEG_MIN=1"
env $CH timeout -k 10 300 python tests/long/fuzz_chain.py 300 3 20000 > gpurun_out/r04t_fuzz_chain.log 2>&1; echo "fuzz_chain forced rc=$?"; tail -n 2 gpurun_out/r04t_fuzz_chain.log
timeout -k 10 300 python tests/long/fuzz_chain.py 200 4 200000 > gpurun_out/r04t_fuzz_chain_dflt.log 2>&1; echo "fuzz_chain default rc=$?"; tail -n 2 gpurun_out/r04t_fuzz_chain_dflt.log
timeout -k 10 300 python tests/long/fuzz_raw.py 500 41 > gpurun_out/r04t_fuzz_raw.log 2>&1; echo "fuzz_raw rc=$?"; tail -n 2 gpurun_out/r04t_fuzz_raw.log
timeout -k 10 600 python -m pytest tests/test_gpu_classes_digest.py tests/test_gpu_container_fused.py -x -q > gpurun_out/r04t_digest.log 2>&1; echo "digest rc=$?"; tail -n 3 gpurun_out/r04t_digest.log
TC_SA_TRACE=1 timeout -k 10 300 python scripts/classes_bench.py 268435456 repeat_4KiB,repeat_1MiB > gpurun_out/r04t_periodic_28.txt 2> gpurun_out/r04t_periodic_28.err; echo "2^28 rc=$?"; cut -c1-260 gpurun_out/r04t_periodic_28.txt
timeout -k 10 300 python scripts/classes_bench.py 1073741824 repeat_4KiB,repeat_1MiB,genome_like,acgtn > gpurun_out/r04t_periodic_30.txt 2> gpurun_out/r04t_periodic_30.err; echo "2^30 rc=$?"; cut -c1-260 gpurun_out/r04t_periodic_30.txt
grep -E "chain|round|ranks" gpurun_out/r04t_periodic_28.err | head -40
