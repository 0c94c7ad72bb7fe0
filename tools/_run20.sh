#!/bin/bash
# round-3 standing of every benched workload (resident + end to end), after the piecewise group kernel
O=gpurun_out/r3q; mkdir -p $O
python bench.py > $O/ns.json 2> $O/ns.err
HIPSTR_SYNTH_IMPERFECT=1.0 python bench.py --no-cpu-baseline --no-pipeline > $O/ns_imp.json 2> $O/ns_imp.err
for w in c5 p30 c4; do python bench.py --workload $w --no-cpu-baseline > $O/$w.json 2> $O/$w.err; done
python - <<'PY'
import json
for n in ("ns","ns_imp","c5","p30","c4"):
    try:
        d=json.loads([l for l in open("gpurun_out/r3q/%s.json"%n) if l.startswith("{")][-1])
    except Exception as e:
        print(n, "FAILED", e); continue
    e=d.get("end_to_end") or {}
    print(n, "value", round(d["value"]/1e6,2), "ms/step", round(d["ms_per_step"],2), "phases", {k:round(v,1) for k,v in d["roofline"]["phase_ms"].items()}, "e2e", round(e.get("alignments_per_s",0)/1e6,1), e.get("fraction_of_resident_rate"))
PY
