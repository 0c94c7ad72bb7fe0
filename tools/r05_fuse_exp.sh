#!/bin/bash
# where the fused trailing-flank kernel's time goes: HS_FUSE_EXP=3 build prints s_memtime per stage for a few workgroups; then the product build, fused on / off
mkdir -p gpurun_out/r05
{
HIPSTR_HMM_LIB=$PWD/hipstr_amd/csrc/ablate/libhipstr_hmm_fx3.so timeout 600 python bench.py --workload ns --steps 1 --warmup 0 --no-cpu-baseline --no-pipeline 2>&1 | grep -E "fused wg" | head -20
for wl in ns p30; do for f in 1 0; do
  echo "== $wl fused $f"
  HIPSTR_TRAIL_FUSED=$f timeout 600 python bench.py --workload $wl --steps 5 --no-cpu-baseline --no-pipeline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), d['roofline']['phase_ms'])"
done; done
} > gpurun_out/r05/fuse_exp.txt 2>&1
cat gpurun_out/r05/fuse_exp.txt
