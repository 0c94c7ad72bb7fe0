#!/bin/bash
# flow_profile.sh — where one host thread of the reference's genotype() goes when the MI355X core is underneath (run on the GPU box):
#   tools/flow_profile.sh [out_dir]     -> <out_dir>/flow_profile.txt (buckets), flow_rates.txt (loci/s, threads x stream)
# libflow_mi355x.so = the caller UNEDITED (include switch + posterior body); _batched = + the optional retrace_alignments body; _nw = + EM and NW bodies.
O=${1:-gpurun_out/flow}; mkdir -p $O
L=oracle/_ref/flow_launcher
for lib in libflow_mi355x.so libflow_mi355x_batched.so libflow_mi355x_nw.so; do
  echo "== $lib --loci 96 --threads 1 --profile" >> $O/flow_profile.txt
  $L oracle/_ref/$lib --loci 96 --seed 100 --threads 1 --profile 2>> $O/flow_profile.txt >> $O/flow_profile.txt
done
echo "== libflow_ref.so (CPU) --loci 96 --threads 1 / 16" >> $O/flow_rates.txt
$L oracle/_ref/libflow_ref.so --loci 96 --seed 100 --threads 1 --profile >> $O/flow_rates.txt 2>&1
$L oracle/_ref/libflow_ref.so --loci 96 --seed 100 --threads 16 >> $O/flow_rates.txt 2>&1
# the CPU reference on the SAME loci as the MI355X rate rows below (16 threads: 768 loci take ~3 s): every rate row has a CPU digest beside it
echo "== libflow_ref.so (CPU) --loci ${FLOW_RATE_LOCI:-768} --threads 16" >> $O/flow_rates.txt
$L oracle/_ref/libflow_ref.so --loci ${FLOW_RATE_LOCI:-768} --seed 100 --threads 16 >> $O/flow_rates.txt 2>&1
for lib in libflow_mi355x.so libflow_mi355x_batched.so; do
for t in 1 4 16; do
  echo "== $lib --threads $t" >> $O/flow_rates.txt
  $L oracle/_ref/$lib --loci ${FLOW_RATE_LOCI:-768} --seed 100 --threads $t >> $O/flow_rates.txt 2>&1
  echo "== $lib --threads $t --stream" >> $O/flow_rates.txt
  $L oracle/_ref/$lib --loci ${FLOW_RATE_LOCI:-768} --seed 100 --threads $t --stream >> $O/flow_rates.txt 2>&1
done
done
echo "== libflow_mi355x.so --threads 1 / 16, prefetch off (HIPSTR_ADAPTER_PREFETCH=0: one device call per traced read, as in round 3)" >> $O/flow_rates.txt
HIPSTR_ADAPTER_PREFETCH=0 $L oracle/_ref/libflow_mi355x.so --loci 96 --seed 100 --threads 1 >> $O/flow_rates.txt 2>&1
HIPSTR_ADAPTER_PREFETCH=0 $L oracle/_ref/libflow_mi355x.so --loci 96 --seed 100 --threads 16 >> $O/flow_rates.txt 2>&1
cat $O/flow_profile.txt $O/flow_rates.txt
