#!/bin/bash
HIPSTR_TRACE_TIMING=1 oracle/_ref/flow_launcher oracle/_ref/libflow_mi355x_batched.so --loci 6 --seed 100 --threads 1 2>&1 | grep -i "hipstr_hmm_trace" | tail -12 > gpurun_out/flow_timing.txt
cat gpurun_out/flow_timing.txt
