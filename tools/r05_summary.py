#!/usr/bin/env python
"""Key numbers of a tools/r05_final.sh run (gpurun_out/r05/, gpurun_out/flow/, gpurun_out/fuzz_big.*), for DESIGN.md / README.md."""
import glob, json, os, re, sys
R = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
def last_json(f):
    return json.loads(open(f).read().strip().splitlines()[-1])
for f in sorted(glob.glob(os.path.join(R, "r05", "r05_bench_*.json"))):
    d = last_json(f)
    e = d.get("end_to_end") or {}
    print(os.path.basename(f), "value %.2f M/s" % (d["value"]/1e6), "ms/step %.2f" % d["ms_per_step"],
          "e2e %.1f" % (d.get("value_end_to_end", 0)/1e6) if d.get("value_end_to_end") else "",
          "e2e@8gpu-share %.1f" % (d.get("value_end_to_end_host_share_8gpu", 0)/1e6) if d.get("value_end_to_end_host_share_8gpu") else "",
          "phases", {k: round(v, 2) for k, v in d["roofline"].get("phase_ms", {}).items()})
    if "roofline" in d and os.path.basename(f) == "r05_bench_ns.json":
        print("   roofline", {k: d["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic") if k in d["roofline"]})
        print("   cpu_baseline", d.get("cpu_baseline"))
        print("   valu", {k: round(v.get("fp64_frac_of_peak", 0), 3) for k, v in (d.get("valu") or {}).get("phases", {}).items()}, (d.get("valu") or {}).get("pass_fp64_frac_of_peak"))
        print("   latency", d.get("one_locus_process_reads_latency"))
        print("   pipeline", json.dumps(d.get("pipeline"))[:900])
    if os.path.basename(f) in ("r05_bench_c3.json", "r05_bench_c4.json"):
        print("   extra", json.dumps({k: v for k, v in d.items() if k.startswith("c3") or k.startswith("c4")})[:600])
for f in sorted(glob.glob(os.path.join(R, "r05", "r05_e2e_*.json"))):
    d = last_json(f); e = d.get("end_to_end", d)
    print(os.path.basename(f), "%.1f M/s" % (e["alignments_per_s"]/1e6), "cpu us/locus %.1f" % e["process_cpu_us_per_locus"], "batches", e["batches"], {k: round(v, 3) for k, v in e.get("cpu_seconds_by_role", {}).items()})
for f in sorted(glob.glob(os.path.join(R, "r05", "r05_*_kernel_stats.txt"))):
    print("==", os.path.basename(f)); print("".join(open(f).readlines()[1:12]))
for f in sorted(glob.glob(os.path.join(R, "r05", "r05_*_pmc_traffic.json"))):
    d = json.load(open(f)); print(os.path.basename(f), {k: d[k] for k in d if not isinstance(d[k], (dict, list))})
p = os.path.join(R, "flow", "flow_rates.txt")
if os.path.exists(p):
    cur = None
    for l in open(p):
        if l.startswith("=="): cur = l.strip()
        elif l.startswith("{"): d = json.loads(l); print(cur, "threads", d["threads"], "stream", d["stream"], "%.1f loci/s (%.1f without the driver's own work)" % (d["loci_per_s"], d["genotype_loci_per_s"]))
for f in (os.path.join(R, "fuzz_big.log"),):
    if os.path.exists(f): print(open(f).read().strip().splitlines()[-1])
