#!/bin/bash
# chain rounds over text lengths / periods (scripts/dbg/chain_probe.py) with the tables traced; the two periodic classes at 1 GiB
mkdir -p gpurun_out
TC_SA_TRACE=1 timeout -k 10 300 python scripts/dbg/chain_probe.py > gpurun_out/r04x_chain_probe.txt 2> gpurun_out/r04x_chain_probe.err; cat gpurun_out/r04x_chain_probe.txt; grep "chain tables" gpurun_out/r04x_chain_probe.err
timeout -k 10 300 python scripts/classes_bench.py 1073741824 repeat_4KiB,repeat_1MiB 2>/dev/null | cut -c1-300
