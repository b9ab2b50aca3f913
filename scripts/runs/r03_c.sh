#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
echo "--- plain (no profiler), hipMalloc workspace"; timeout -k 10 200 python scripts/dbg/mode_place.py 6 2>&1 | grep PLACEMENTS
for k in 21 26 30 33; do
  echo "--- VMM workspace, chunks of 2^$k"; TC_WS_VMM=$k TC_BENCH_PLACE=0 timeout -k 10 200 python bench.py --steps 5 --no-cpu-baseline --no-fm 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['container']['ms_per_step_with_container'])"
done
timeout -k 10 1000 bash scripts/mode_pmc.sh \
  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
  "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
  "TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_LEVEL_sum" \
  "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
  "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" 2>&1 | tee gpurun_out/r03c_mode_pmc.txt
