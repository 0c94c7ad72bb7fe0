#!/usr/bin/env python
"""HBM traffic per kernel from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), as MI355X_MICROARCH.md §HBM
prescribes: FETCH_SIZE/WRITE_SIZE are KB per dispatch derived from TCC_EA0_RDREQ/WRREQ; on gfx950 FETCH_SIZE counts a
128-B read request of a wide coalesced stream as 64 B, so the read side is reported raw and x2 (upper bound); WRITE_SIZE is
uncalibrated and reported raw.
usage: python tools/pmc_traffic.py <fetch.db> <write.db> <alignments_per_launch> ["<command>"] > profiles/<name>.json"""
import json
import sqlite3
import sys


import re


def passes_of(command):
    m1, m2 = re.search(r"--steps\s+(\d+)", command), re.search(r"--warmup\s+(\d+)", command)
    return max(1, (int(m1.group(1)) if m1 else 1) + (int(m2.group(1)) if m2 else 0))


def per_kernel(db, counter):
    """Bytes per kernel and PASS: the average over its dispatches times its dispatches per pass (a chunked pass launches a kernel per chunk)."""
    out = {}
    passes = passes_of(sys.argv[4] if len(sys.argv) > 4 else "")
    for name, n, avg in sqlite3.connect(db).execute(
            "select kernel_name,count(distinct dispatch_id),avg(value) from counters_collection where counter_name=? group by kernel_name", (counter,)):
        if "hs_" in name:
            key = name.split("::")[-1].split("(")[0]
            out[key] = out.get(key, 0.0) + avg * 1024.0 * n / passes
    return out


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
n_aln = float(sys.argv[3])
res = {"command": sys.argv[4] if len(sys.argv) > 4 else "", "alignments_per_launch": n_aln, "kernels": {}}        # bench.py matches the workload by the command
tot_f = tot_w = 0.0
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    tot_f += f; tot_w += w
    res["kernels"][k] = {"fetch_bytes_per_launch_raw": f, "write_bytes_per_launch_raw": w}
res["pass"] = {"fetch_bytes_raw": tot_f, "fetch_bytes_x2": 2 * tot_f, "write_bytes_raw": tot_w,
               "bytes_per_alignment_raw": (tot_f + tot_w) / n_aln, "bytes_per_alignment_fetch_x2": (2 * tot_f + tot_w) / n_aln}
print(json.dumps(res, indent=1))
