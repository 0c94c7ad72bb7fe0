#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats) into a small text table for profiles/.
usage: python tools/rocprof_summary.py <results.db> [bench.log] > profiles/<name>.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
print("# rocprofv3 --kernel-trace --stats summary of %s" % sys.argv[1])
print("%-40s %8s %14s %14s %8s" % ("kernel", "calls", "total_ms", "avg_ms", "pct"))
rows = list(db.execute("select name,count(*),sum(end-start),avg(end-start) from kernels group by name order by sum(end-start) desc"))
grand = sum(r[2] for r in rows) or 1
for name, calls, total, avg in rows:      # start/end are nanoseconds
    print("%-40s %8d %14.3f %14.3f %8.2f" % (name[:40], calls, total / 1e6, avg / 1e6, 100.0 * total / grand))
print()
print("%-40s %6s %6s %6s %8s %8s %s" % ("kernel", "vgpr", "agpr", "sgpr", "lds", "scratch", "grid x wg (first dispatch)"))
seen = set()
for r in db.execute("select name,vgpr_count,accum_vgpr_count,sgpr_count,lds_size,scratch_size,grid_x,grid_y,grid_z,workgroup_x from kernels order by start"):
    if r[0] in seen:
        continue
    seen.add(r[0])
    print("%-40s %6d %6d %6d %8d %8d %dx%dx%d / %d" % (r[0][:40], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9]))
if len(sys.argv) > 2:
    print()
    print("# bench.py line of the same command")
    for line in open(sys.argv[2]):
        if line.startswith("{"):
            print(line.strip())
