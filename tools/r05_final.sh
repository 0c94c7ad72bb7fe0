#!/bin/bash
# r05_final.sh — everything the round's final artefacts come from, in one gpurun call: profiles + bench lines (tools/r05_profiles.sh), the
# flow profile / rates (with the 768-locus CPU row), the big fuzz, the boundary sweeps.  Results under gpurun_out/r05/ (+ flow/, fuzz_big.txt).
uptime > gpurun_out/r05_uptime.txt
bash tools/r05_profiles.sh > gpurun_out/r05_profiles.log 2>&1
rm -rf gpurun_out/flow; bash tools/flow_profile.sh gpurun_out/flow > gpurun_out/flow.log 2>&1
bash tools/fuzz_big.sh 12 50 gpurun_out/fuzz_big.txt > gpurun_out/fuzz_big.log 2>&1
bash tools/fuzz_edges.sh gpurun_out/fuzz_boundaries.txt > gpurun_out/fuzz_edges.log 2>&1
tail -3 gpurun_out/fuzz_big.txt
uptime >> gpurun_out/r05_uptime.txt
ls gpurun_out/r05 | head -80
