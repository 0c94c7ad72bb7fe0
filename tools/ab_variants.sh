for v in "$@"; do
  L=hipstr_amd/csrc/ablate/libhipstr_hmm_$v.so
  HIPSTR_HMM_LIB=$PWD/$L python bench.py --loci 400 --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,2), round(d['roofline']['pass_ms'],2), {k:round(v,2) for k,v in d['roofline']['phase_ms'].items()})"
done
