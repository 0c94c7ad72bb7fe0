#!/bin/bash
# Interrupted repeats: resident rates of the NS shape by generator mode (random substitution in every alt allele; 1/2/3 interruptions inherited from the reference allele)
out=gpurun_out/${1:-r04_imp}; mkdir -p $out
run(){ name=$1; shift; env "$@" timeout 900 python bench.py --workload ns --loci 400 --no-cpu-baseline --no-pipeline --steps 5 > $out/$name.json 2> $out/$name.err; }
run ns_default HIPSTR_X=0
run ns_imperfect1.0 HIPSTR_SYNTH_IMPERFECT=1.0
run ns_inherit1 HIPSTR_SYNTH_INHERIT=1
run ns_inherit2 HIPSTR_SYNTH_INHERIT=2
run ns_inherit3 HIPSTR_SYNTH_INHERIT=3
