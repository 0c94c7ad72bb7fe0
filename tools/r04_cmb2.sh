#!/bin/bash
# combine kernel variants on one box (libraries hipstr_amd/csrc/libhipstr_hmm_<variant>.so)
out=gpurun_out/${1:-r04_cmb2}; mkdir -p $out; shift
for rep in 1 2; do
  for v in "$@"; do
    L=hipstr_amd/csrc/libhipstr_hmm_$v.so; [ $v = new ] && L=hipstr_amd/csrc/libhipstr_hmm.so
    HIPSTR_HMM_LIB=$PWD/$L timeout 900 python bench.py --no-cpu-baseline --no-pipeline --steps 10 > $out/ns_${v}_$rep.json 2> $out/ns_${v}_$rep.err
  done
done
python - $out <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        ph = (d.get("roofline") or {}).get("phase_ms") or {}
        print(os.path.basename(f), round(d["value"]/1e6, 2), round(d["ms_per_step"], 2), {k: round(v, 2) for k, v in ph.items()})
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
