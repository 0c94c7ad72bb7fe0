#!/bin/bash
# piecewise group kernel: fuzz + A/B
mkdir -p gpurun_out
timeout 900 python tools/fuzz_align.py 80 777 > gpurun_out/fuzz_pw.txt 2>&1; tail -5 gpurun_out/fuzz_pw.txt
HIPSTR_DEBUG_REDO=3 timeout 600 python tools/fuzz_align.py 30 778 > gpurun_out/fuzz_pw_redo.txt 2>&1; tail -3 gpurun_out/fuzz_pw_redo.txt
rm -f gpurun_out/ab_pw.txt
tools/gpu_ab.sh pw X=1 HIPSTR_STR_GROUP_PW=0 HIPSTR_SYNTH_IMPERFECT=1.0 HIPSTR_SYNTH_IMPERFECT=1.0,HIPSTR_STR_GROUP_PW=0
