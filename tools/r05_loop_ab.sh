#!/bin/bash
# A/B of library variants (tools/build_variant.sh <name> -D...) on the interrupted-repeat modes and the default mix (NS shape, 400 loci):
#   tools/r05_loop_ab.sh <out dir under gpurun_out> <variant or "product">...      AB_REPS (2), AB_LOCI (400), AB_MODES
out=gpurun_out/$1; shift; mkdir -p $out
run(){ lib=$1; name=$2; shift 2; L=""; [ "$lib" != product ] && L="HIPSTR_HMM_LIB=$PWD/hipstr_amd/csrc/ablate/libhipstr_hmm_$lib.so"
  env "$@" $L timeout 900 python bench.py --workload ns --loci ${AB_LOCI:-400} --no-cpu-baseline --no-pipeline --steps 5 2> $out/$name.$lib.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$name', round(d['value']/1e6,2), 'M/s', {k:round(v,2) for k,v in d['roofline']['phase_ms'].items()})"; }
MODES=${AB_MODES:-"default imperfect inherit1 inherit2 inherit3"}
for rep in $(seq 1 ${AB_REPS:-2}); do
for lib in "$@"; do
  for m in $MODES; do
    case $m in
      default) run $lib default A=1;;
      imperfect) run $lib imperfect HIPSTR_SYNTH_IMPERFECT=1.0;;
      inherit1) run $lib inherit1 HIPSTR_SYNTH_INHERIT=1;;
      inherit2) run $lib inherit2 HIPSTR_SYNTH_INHERIT=2;;
      inherit3) run $lib inherit3 HIPSTR_SYNTH_INHERIT=3;;
    esac
  done
done
done 2>&1 | tee $out/ab.txt
