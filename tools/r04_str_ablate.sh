#!/bin/bash
# STR phase of the interrupted-repeat modes with parts of the grouped kernels left out (timing builds: results invalid): tools/r04_str_ablate.sh <out>
out=gpurun_out/$1; mkdir -p $out
run(){ name=$1; lib=$2; shift 2; env "$@" ${lib:+HIPSTR_HMM_LIB=$lib} timeout 900 python bench.py --workload ns --loci 400 --no-cpu-baseline --no-pipeline --steps 3 2> "$out/${name// /_}.err" | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M/s', {k:round(v,2) for k,v in d['roofline']['phase_ms'].items()})"; }
for mode in HIPSTR_SYNTH_INHERIT=2 HIPSTR_SYNTH_IMPERFECT=1.0 HIPSTR_SYNTH_INHERIT=1; do
  run "$mode full" "" $mode
  run "$mode no-read-end-sums" hipstr_amd/csrc/ablate/libhipstr_hmm_gabl1.so $mode
  run "$mode no-evaluation" hipstr_amd/csrc/ablate/libhipstr_hmm_gabl2.so $mode
  run "$mode no-table-phase" hipstr_amd/csrc/ablate/libhipstr_hmm_gabl3.so $mode
done
