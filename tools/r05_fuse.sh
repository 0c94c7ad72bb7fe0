#!/bin/bash
# trailing flanks + compute_aln_logprob as one item (HIPSTR_TRAIL_FUSED, default on) against the two plain items + hs_combine_kernel
mkdir -p gpurun_out/r05
{
timeout 1500 python -m pytest tests/test_hmm_gpu.py -m gpu -x -q 2>&1 | tail -5
for wl in ns p30; do for f in 1 0 1; do
  echo "== $wl fused $f"
  HIPSTR_TRAIL_FUSED=$f timeout 600 python bench.py --workload $wl --steps 5 --no-cpu-baseline --no-pipeline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), d['roofline']['phase_ms'])"
done; done
} > gpurun_out/r05/fuse.txt 2>&1
tail -40 gpurun_out/r05/fuse.txt
