#!/bin/bash
# round 4, second session: full -m gpu suite on the chain-round build, then the per-step trace of three classes at 1 GiB
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04u_gputests.log 2>&1
echo "gpu tests rc=$?"; tail -n 3 gpurun_out/r04u_gputests.log
for c in genome_like zipf_words runs_p0.9; do
  TC_SA_TRACE=1 timeout -k 10 300 python scripts/classes_bench.py 1073741824 $c > gpurun_out/r04u_trace_$c.txt 2> gpurun_out/r04u_trace_$c.err; echo "$c rc=$?"; cut -c1-200 gpurun_out/r04u_trace_$c.txt
done
