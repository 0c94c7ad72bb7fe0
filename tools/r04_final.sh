#!/bin/bash
# r04_final.sh — everything the round's final artefacts come from, in one gpurun call: profiles + bench lines (tools/r04_profiles.sh), the
# flow profile / rates, the big fuzz.  Results under gpurun_out/r04/ (+ flow/, fuzz_big.txt).
uptime > gpurun_out/r04_uptime.txt
bash tools/r04_profiles.sh > gpurun_out/r04_profiles.log 2>&1
rm -rf gpurun_out/flow; bash tools/flow_profile.sh gpurun_out/flow > gpurun_out/flow.log 2>&1
bash tools/fuzz_big.sh 12 50 gpurun_out/fuzz_big.txt > gpurun_out/fuzz_big.log 2>&1
tail -3 gpurun_out/fuzz_big.txt
uptime >> gpurun_out/r04_uptime.txt
ls gpurun_out/r04 | head -60
