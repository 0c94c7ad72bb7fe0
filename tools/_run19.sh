#!/bin/bash
export HIPSTR_SYNTH_IMPERFECT=1.0
HIPSTR_HMM_LIB=hipstr_amd/csrc/ablate/libhipstr_hmm_gt.so python bench.py --loci 400 --steps 1 --warmup 0 --no-cpu-baseline --no-pipeline 2>&1 | grep "^grp" | head -60 > gpurun_out/gt_pw.txt
cat gpurun_out/gt_pw.txt | head -40
