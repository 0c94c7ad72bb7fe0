#!/bin/bash
# combine kernel A/B on one box: previous library vs the new one; parity subset first
out=gpurun_out/${1:-r04_cmb}; mkdir -p $out
./tools/swap_probe > $out/swap_probe.txt 2>&1
timeout 1500 python -m pytest tests/test_hmm_gpu.py tests/test_limits_gpu.py tests/test_seeded.py -x -q -m gpu 2>&1 | tail -5 > $out/tests.txt
for rep in 1 2; do
  for lib in prev new; do
    L=hipstr_amd/csrc/libhipstr_hmm.so; [ $lib = prev ] && L=hipstr_amd/csrc/libhipstr_hmm_prev.so
    HIPSTR_HMM_LIB=$PWD/$L timeout 900 python bench.py --no-cpu-baseline --no-pipeline --steps 10 > $out/ns_${lib}_$rep.json 2> $out/ns_${lib}_$rep.err
  done
done
for lib in prev new; do
  L=hipstr_amd/csrc/libhipstr_hmm.so; [ $lib = prev ] && L=hipstr_amd/csrc/libhipstr_hmm_prev.so
  for wl in p30 c2 c5; do
    HIPSTR_HMM_LIB=$PWD/$L timeout 900 python bench.py --workload $wl --no-cpu-baseline --no-pipeline --steps 10 > $out/${wl}_${lib}.json 2> $out/${wl}_${lib}.err
  done
done
cat $out/swap_probe.txt; cat $out/tests.txt
python - $out <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        ph = (d.get("roofline") or {}).get("phase_ms") or {}
        print(os.path.basename(f), d["value"], d["ms_per_step"], ph)
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
