#!/bin/bash
tools/r03_profiles.sh > gpurun_out/r03_profiles.log 2>&1
tools/fuzz_big.sh 12 60 gpurun_out/r03/r03_fuzz.txt 2>&1 | tail -2
timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
