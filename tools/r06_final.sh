#!/bin/bash
# r06_final.sh — everything the round's final artefacts come from, in one gpurun call: the GPU suite, smoke(), profiles + bench lines
# (tools/r06_profiles.sh), the flow profile / rates, the fuzzers (incl. the round's "big" mode).  Results under gpurun_out/r06/ (+ flow/).
mkdir -p gpurun_out/r06
uptime > gpurun_out/r06/r06_uptime.txt
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r06/r06_pytest_gpu.txt 2>&1; tail -5 gpurun_out/r06/r06_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06/r06_smoke.txt 2>&1; tail -1 gpurun_out/r06/r06_smoke.txt
timeout 3000 bash tools/r06_profiles.sh > gpurun_out/r06_profiles.log 2>&1
rm -rf gpurun_out/flow; timeout 900 bash tools/flow_profile.sh gpurun_out/flow > gpurun_out/flow.log 2>&1
timeout 900 bash tools/fuzz_big.sh 12 50 gpurun_out/r06/r06_fuzz.txt > gpurun_out/fuzz_big.log 2>&1
timeout 600 bash tools/fuzz_edges.sh gpurun_out/r06/r06_fuzz_boundaries.txt > gpurun_out/fuzz_edges.log 2>&1
for i in 1 2 3 4; do timeout 400 python tools/fuzz_align.py 40 $((9100+i)) big > /tmp/fb_$i.txt 2>&1 & done; wait
for i in 1 2 3 4; do echo "forward big $i: $(tail -n 1 /tmp/fb_$i.txt)"; grep -h "MISMATCH" /tmp/fb_$i.txt | head -3; done > gpurun_out/r06/r06_fuzz_sizes.txt
HIPSTR_FUZZ_BIG=1 timeout 400 python tools/fuzz_post.py 24 9200 >> gpurun_out/r06/r06_fuzz_sizes.txt 2>&1
tail -3 gpurun_out/r06/r06_fuzz.txt; cat gpurun_out/r06/r06_fuzz_sizes.txt | tail -8
uptime >> gpurun_out/r06/r06_uptime.txt
ls gpurun_out/r06 | head -80
