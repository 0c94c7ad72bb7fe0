#!/usr/bin/env python
"""Prints one line per end-to-end JSON (bench.py --e2e-only) in a directory: rate, CPU per locus, CPU by role."""
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        roles = d.get("cpu_seconds_by_role", {})
        n = d["submissions"]
        print("%-28s %6.1f M/s %7.1f ms/pass cpu %5.1f us/locus [%s] wait %.2fs thr %s cpus %s batches %s" % (
            os.path.basename(f), d["alignments_per_s"] / 1e6, d["ms_per_pass"], d.get("process_cpu_us_per_locus", 0),
            " ".join("%s %.1f" % (k, 1e6 * v / n) for k, v in roles.items()), d["collector_wait_seconds"], d.get("host_threads"), d.get("cpus_allowed"), d["batches"]))
    except Exception as ex:
        print(f, "ERR", ex)
