#!/bin/bash
# A/B of library variants on one box: tools/r06_ab.sh <tag> <workload> <loci or 0> <variant> ...   (variant = name under hipstr_amd/csrc/ablate, or "product")
TAG=$1; WL=$2; LOCI=$3; shift 3
mkdir -p gpurun_out
O=gpurun_out/ab_$TAG.txt
for v in "$@"; do
  if [ "$v" = product ]; then unset HIPSTR_HMM_LIB; else export HIPSTR_HMM_LIB=$PWD/hipstr_amd/csrc/ablate/libhipstr_hmm_$v.so; fi
  L=""; [ "$LOCI" != 0 ] && L="--loci $LOCI"
  echo "== $WL $v" >> $O
  timeout 240 python bench.py --workload $WL $L --steps 5 --warmup 1 --no-cpu-baseline --no-pipeline 2>gpurun_out/ab_${TAG}_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M/s pass_ms', round(d['roofline']['pass_ms'],2), {k:round(v,2) for k,v in d['roofline']['phase_ms'].items()})" >> $O 2>&1
done
unset HIPSTR_HMM_LIB
cat $O
