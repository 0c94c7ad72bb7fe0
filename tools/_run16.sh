#!/bin/bash
# piecewise group kernel: fuzz + A/B
mkdir -p gpurun_out
timeout 900 python tools/fuzz_align.py ${FUZZ_N:-40} 779 > gpurun_out/fuzz_pw.txt 2>&1; tail -3 gpurun_out/fuzz_pw.txt
rm -f gpurun_out/ab_pw.txt
tools/gpu_ab.sh pw X=1 HIPSTR_SYNTH_IMPERFECT=1.0
