#!/usr/bin/env python
"""Per-kernel SQ instruction counters from one rocprofv3 --pmc pass (own run, --kernel-trace only) -> profiles/<name>_sq_counters.json.
VALU issue utilisation = SQ_INSTS_VALU x 4 cycles per wave64 instruction / (1024 SIMDs x 2.4 GHz x kernel duration); counters and
durations are averages per dispatch.
usage: python tools/sq_counters.py <sq_results.db> <alignments_per_launch> "<command line of the pass>" > profiles/<name>.json"""
import json
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n_aln = float(sys.argv[2])
CLOCK, SIMDS, CYC = 2.4e9, 1024, 4
names = {"SQ_INSTS_VALU": "valu_insts", "SQ_INSTS_SALU": "salu_insts", "SQ_INSTS_LDS": "lds_insts", "SQ_WAVE_CYCLES": "wave_cycles",
         "SQ_WAIT_INST_ANY": "wait_inst_any"}
kern = {}
for name, counter, avg in db.execute("select kernel_name,counter_name,avg(value) from counters_collection group by kernel_name,counter_name"):
    if "hs_" not in name or counter not in names:
        continue
    key = name.split("::")[-1].split("(")[0]
    kern.setdefault(key, {})[names[counter]] = avg
for name, dur in db.execute("select kernel_name, avg(duration) from (select distinct dispatch_id, kernel_name, duration from counters_collection) group by kernel_name"):
    key = name.split("::")[-1].split("(")[0]
    if key in kern:
        kern[key]["duration_ns"] = dur
        kern[key]["valu_issue_utilisation"] = kern[key]["valu_insts"] * CYC / (SIMDS * CLOCK * dur * 1e-9)
print(json.dumps({"command": sys.argv[3], "alignments_per_launch": n_aln, "clock_hz_assumed": CLOCK, "simds": SIMDS,
                  "cycles_per_wave64_valu_inst": CYC, "kernels": dict(sorted(kern.items()))}, indent=1))
