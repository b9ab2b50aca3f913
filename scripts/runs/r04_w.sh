#!/bin/bash
# where iid ACGT's 33 ms go (kernel stats), and what its levels / big finish cost with keys only (forced: the attempt is
# thrown away -- 261 586 ties exceed the table -- but its kernels show in the trace)
mkdir -p gpurun_out
bash scripts/prof_class.sh acgt4 1073741824 > gpurun_out/r04w_acgt4_kernels.txt 2>&1; cat gpurun_out/r04w_acgt4_kernels.txt
mv gpurun_out/prof_cls_acgt4 gpurun_out/prof_cls_acgt4_vals
TC_SA_MSD_KEYONLY=2 bash scripts/prof_class.sh acgt4 1073741824 > gpurun_out/r04w_acgt4_keyonly_kernels.txt 2>&1; cat gpurun_out/r04w_acgt4_keyonly_kernels.txt
rm -rf gpurun_out/prof_cls_acgt4 gpurun_out/prof_cls_acgt4_vals
