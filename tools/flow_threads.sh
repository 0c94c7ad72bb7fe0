#!/bin/bash
# flow_threads.sh — the reference's genotype() over the device with more host threads than cores (threads mostly wait for the device):
#   tools/flow_threads.sh [out_dir]
O=${1:-gpurun_out/flowt}; mkdir -p $O
L=oracle/_ref/flow_launcher
for lib in libflow_mi355x.so libflow_mi355x_batched.so; do
for t in ${FLOW_THREADS:-16 32 64 128}; do
  for s in "" "--stream"; do
    echo "== $lib --threads $t $s" >> $O/flow_threads.txt
    timeout 300 $L oracle/_ref/$lib --loci ${FLOW_LOCI:-768} --seed 100 --threads $t $s >> $O/flow_threads.txt 2>&1
  done
done
done
cat $O/flow_threads.txt
