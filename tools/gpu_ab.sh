#!/bin/bash
# A/B timing of library variants on the GPU box: tools/gpu_ab.sh <tag> [ENV=VAL,...]...   each argument = one run of the NS bench at 400 loci
TAG=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  envs=$(echo "$v" | tr ',' ' ')
  echo "== $v" >> gpurun_out/ab_$TAG.txt
  env $envs python bench.py --loci ${AB_LOCI:-400} --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline 2>gpurun_out/ab_${TAG}_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M/s pass_ms', round(d['roofline']['pass_ms'],2), {k:round(v,2) for k,v in d['roofline']['phase_ms'].items()})" >> gpurun_out/ab_$TAG.txt 2>&1
done
cat gpurun_out/ab_$TAG.txt
