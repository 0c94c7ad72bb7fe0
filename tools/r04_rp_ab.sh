#!/bin/bash
# A/B of hs_str_group_kernel_rp builds on the interrupted-repeat modes (NS shape, 400 loci): tools/r04_rp_ab.sh <out> <lib or ""> 
out=gpurun_out/$1; mkdir -p $out
LIB=$2
run(){ name=$1; shift; env "$@" ${LIB:+HIPSTR_HMM_LIB=$LIB} timeout 900 python bench.py --workload ns --loci 400 --no-cpu-baseline --no-pipeline --steps 5 2> $out/$name.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M/s', {k:round(v,2) for k,v in d['roofline']['phase_ms'].items()})"; }
run inherit2 HIPSTR_SYNTH_INHERIT=2
run inherit3 HIPSTR_SYNTH_INHERIT=3
run inherit2 HIPSTR_SYNTH_INHERIT=2
run inherit3 HIPSTR_SYNTH_INHERIT=3
