#!/bin/bash
# stutter EM + posterior A/B: tools/r06_em_ab.sh <variant> ...  (variant = name under hipstr_amd/csrc/ablate, or "product"); c3 shape, 3000 loci
for v in "$@"; do
  if [ "$v" = product ]; then unset HIPSTR_HMM_LIB; else export HIPSTR_HMM_LIB=$PWD/hipstr_amd/csrc/ablate/libhipstr_hmm_$v.so; fi
  echo "== c3 3000 loci $v"
  timeout 300 python bench.py --workload c3 --loci 3000 --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1), 'ms per step', d['c3_step'])"
  echo "== c4 $v"
  timeout 300 python bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), 'M/s', {k:round(v,5) for k,v in d['c4_step'].items() if isinstance(v,float)})"
done
unset HIPSTR_HMM_LIB
