#!/bin/bash
# round 4: FM count rate against the text size with the pair vectors as mapped chunks (default from 2^31 bytes on) and as one block
echo "# pair vectors as 2^28-byte mapped chunks from 2 GiB on (default)" > gpurun_out/r04_fm_sweep.txt
timeout -k 10 500 python scripts/fm_sweep.py 4000000 28,29,30 >> gpurun_out/r04_fm_sweep.txt 2>&1
echo "# TC_FM_VMM=0: one hipMalloc block" >> gpurun_out/r04_fm_sweep.txt
TC_FM_VMM=0 timeout -k 10 500 python scripts/fm_sweep.py 4000000 29,30 >> gpurun_out/r04_fm_sweep.txt 2>&1
echo "# TC_FM_VMM=30 (1 GiB chunks)" >> gpurun_out/r04_fm_sweep.txt
TC_FM_VMM=30 timeout -k 10 500 python scripts/fm_sweep.py 4000000 30 >> gpurun_out/r04_fm_sweep.txt 2>&1
