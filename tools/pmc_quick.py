#!/usr/bin/env python
"""Per-kernel averages of the counters of one rocprofv3 --pmc pass: python tools/pmc_quick.py <results.db> [name filter]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = {}
for name, counter, avg, n in db.execute("select kernel_name,counter_name,avg(value),count(*) from counters_collection group by kernel_name,counter_name"):
    if flt in name: rows.setdefault(name, {})[counter] = avg
for name, d in db.execute("select kernel_name, avg(duration) from (select distinct dispatch_id, kernel_name, duration from counters_collection) group by kernel_name"):
    if name in rows: rows[name]["duration_us"] = d/1e3
for name, r in rows.items():
    print(name[:60]); print("   ", {k: (round(v, 1) if v < 1e6 else float("%.4g" % v)) for k, v in sorted(r.items())})
