#!/usr/bin/env python
"""DESIGN.md §6's table from the committed artefacts (profiles/r06_*): every number with the file it comes from.  usage: tools/r06_design6.py [dir]"""
import json, os, re, sys
D = sys.argv[1] if len(sys.argv) > 1 else "profiles"
def J(name):
    return json.loads(open(os.path.join(D, name)).read().strip().splitlines()[-1])
def M(x): return "%.1f" % (x / 1e6)
def kstats(name):
    out = {}
    for l in open(os.path.join(D, name)):
        m = re.match(r"(.{40})\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", l)
        if m: out.setdefault(m.group(1).strip(), float(m.group(4)))
    return out
def kfind(ks, key): return sum(v for k, v in ks.items() if key in k)
ns = J("r06_bench_ns.json"); rf = ns["roofline"]; ph = rf["phase_ms"]; v = ns["valu"]["phases"]
ks = kstats("r06_ns_kernel_stats.txt")
tr = json.load(open(os.path.join(D, "r06_ns_pmc_traffic.json")))
rows = []
rows.append(("NS workload (1000 loci × 500 pooled 150 bp reads × 32 alleles = 15.9 M alignments per pass), inputs resident (`value`)",
             "**%s M alignments/s**, %.2f k loci/s, %.1f ms per pass (`r06_bench_ns.json`)" % (M(ns["value"]), ns["loci_per_sec"] / 1e3, ns["ms_per_step"])))
tot = sum(ph.values())
rows.append(("phase split per pass (HIP events on the launch stream, same file)",
             "trailing flank %.1f ms (%.0f %%), STR %.1f (%.0f %%), leading flank incl. column tables %.1f, combine %.1f" % (ph["hs_trail_kernel"], 100 * ph["hs_trail_kernel"] / tot, ph["hs_str_kernel"], 100 * ph["hs_str_kernel"] / tot, ph["hs_lead_kernel"], ph["hs_combine_kernel"])))
rows.append(("rocprofv3 `--kernel-trace --stats` of the same command (`r06_ns_kernel_stats.txt`)",
             "trail %.2f ms; `hs_str_group_kernel_p` %.2f + `_pw` %.2f + `hs_nd_kernel` %.2f + generic %.2f; combine %.2f; lead %.2f + column tables %.2f" % (
                 kfind(ks, "hs_trail_ker"), kfind(ks, "hs_str_group_kernel_p") - kfind(ks, "hs_str_group_kernel_pw"), kfind(ks, "hs_str_group_kernel_pw"), kfind(ks, "hs_nd_kernel"),
                 kfind(ks, "hs_str_kernel_generic"), kfind(ks, "hs_combine_kernel"), kfind(ks, "hs_lead_kern"), kfind(ks, "hs_col_kernel"))))
rows.append(("binding roof: FP64 VALU issue, 39.3 T op/s (`valu` block of the line; counters `%s`, `profile_matches_build: %s`)" % (ns["valu"]["source"], str(ns["valu"]["profile_matches_build"]).lower()),
             "trailing flank **%.2f** of the FP64 peak (VALU pipe %.2f), STR phase %.2f (pipe %.2f), lead %.2f, combine %.2f; pass **%.2f**" % (
                 v["hs_trail_kernel"]["fp64_frac_of_peak"], v["hs_trail_kernel"]["pipe_occupancy_est"], v["hs_str_kernel"]["fp64_frac_of_peak"], v["hs_str_kernel"]["pipe_occupancy_est"],
                 v["hs_lead_kernel"]["fp64_frac_of_peak"], v["hs_combine_kernel"]["fp64_frac_of_peak"], ns["valu"]["pass_fp64_frac_of_peak"])))
rows.append(("HBM \"roofline\" the metric asks for (`roofline` block)",
             "%.1f MB algorithmic per launch (%.2f B per alignment) ÷ %.2f ms = %.2f GB/s = **%.2e** of 8 TB/s; counter traffic of the trailing-flank kernel %.1f GB (FETCH × 2 + WRITE, `%s`), of the pass %.2f KB per alignment raw (%.2f with FETCH × 2) against %.1f B algorithmic" % (
                 rf["algorithmic_bytes_per_launch"] / 1e6, rf["bytes_per_alignment"], rf["kernel_ms"], rf["achieved"], rf["frac"], rf["traffic"] / 1e9, rf["traffic_source"],
                 tr["pass"]["bytes_per_alignment_raw"] / 1e3, tr["pass"]["bytes_per_alignment_fetch_x2"] / 1e3, rf["bytes_per_alignment"])))
e = ns["end_to_end"]; e8 = ns["end_to_end_host_share_8gpu"]; p8 = ns.get("end_to_end_host_share_8gpu_p30", {})
rows.append(("**end to end, host arrays in → results out** (`value_end_to_end`: 1000 single-locus submissions per pass through `hipstr_stream_*`)",
             "**%s M/s = %.2f of resident**; the process pinned to 2 of the node's 16 usable CPUs (`value_end_to_end_host_share_8gpu`) **%s M/s = %.2f**; p30 under the same pinning (`…_p30`) %s M/s = %.2f of its resident rate, %.1f µs of host CPU per locus" % (
                 M(e["alignments_per_s"]), e["fraction_of_resident_rate"], M(e8["alignments_per_s"]), e8["fraction_of_resident_rate"],
                 M(p8.get("alignments_per_s", 0)), p8.get("fraction_of_resident_rate", 0), p8.get("process_cpu_us_per_locus", 0))))
for w, label in (("p30", "production-like shape (`--workload p30`: 4000 loci × 40 reads × 8 alleles, 35-bp flanks)"), ("c2", "configs[1] at literal 30× depth (`--workload c2`: 1000 × 40 × 32)")):
    b = J("r06_bench_%s.json" % w); a = J("r06_e2e_%s_all.json" % w); p = J("r06_e2e_%s_pin2.json" % w)
    ea = a.get("end_to_end", a); ep = p.get("end_to_end", p)
    rows.append((label, "resident %s M/s (`r06_bench_%s.json`); end to end %s M/s = %.2f (`r06_e2e_%s_all.json`), pinned to 2 CPUs %s M/s = %.2f, %.1f µs of host CPU per locus (`r06_e2e_%s_pin2.json`)" % (
        M(b["value"]), w, M(ea["alignments_per_s"]), ea.get("fraction_of_resident_rate_same_process", ea["alignments_per_s"] / b["value"]), w, M(ep["alignments_per_s"]), ep.get("fraction_of_resident_rate_same_process", ep["alignments_per_s"] / b["value"]), ep["process_cpu_us_per_locus"], w)))      # (the fraction: of the same process' resident rate where the file has it)
lat = e["one_locus_process_reads_latency"]
rows.append(("one locus per call (`hipstr_hmm_process_reads` on prepared arrays, median of 200; `one_locus_process_reads_latency` of the NS line)",
             "50 reads × 4 alleles (configs[0]) **%.3f ms**, 40 × 32 **%.3f ms**, 500 × 32 %.3f ms (round 5: 0.142 / 0.184 / 0.406); through the python wrapper incl. the oracle-sized copies 0.20 / 0.26 / 0.52 (`r06_latency.txt`), kernel timeline `r06_lat_trace_align.txt`" % (
                 lat["50x4"]["median_ms"], lat["40x32"]["median_ms"], lat["500x32"]["median_ms"])))
c3 = J("r06_bench_c3.json"); s3 = c3["c3_step"]
rows.append(("configs[2] at its own size (`--workload c3`: 10 000 loci × 600 reads (100 samples × 6) × 32 alleles; a step = stutter EM + forward HMM + posteriors + genotype calls with GL/PL)",
             "**%.2f s per step = %.1f k loci/s, %s M alignments/s**: EM %.2f s (10 000 loci, %d locus-iterations), genotype calls incl. 6.5 GB of GL/PL back to the host %.2f s (`r06_bench_c3.json`; kernels `r06_c3_kernel_stats.txt`)" % (
                 c3["ms_per_step"] / 1e3, c3["loci_per_sec"] / 1e3, M(c3["value"]), s3["stutter_em_s_per_step"], s3["em_iterations"], s3["genotype_calls_s_per_step"])))
c4 = J("r06_bench_c4.json"); c5 = J("r06_bench_c5.json"); s4 = c4["c4_step"]
rows.append(("other shapes", "configs[4] stress (`c5`) %s M/s, step %.1f ms; `--workload c4` (configs[3]: 1000 samples per locus) %s M/s over forward + posteriors + calls (posterior kernel %.2f ms, genotype calls %.1f ms of a %.0f ms step)" % (
    M(c5["value"]), c5["ms_per_step"], M(c4["value"]), 1e3 * s4["posterior_kernel_s"], 1e3 * s4["genotype_calls_s"], c4["ms_per_step"])))
imp = J("r06_bench_ns_imperfect1.0.json"); inh = [J("r06_bench_ns_inherit%d.json" % k) for k in (1, 2, 3)]
rows.append(("interrupted repeats (NS shape, 1000 loci; `r06_bench_ns_{imperfect1.0,inherit1,inherit2,inherit3}.json`; counters `r06_{imperfect,inherit2}_*`)",
             "every alt allele with a random substitution %s M/s (STR phase %.0f ms); 1 / 2 / 3 interruptions inherited from the reference allele **%s / %s / %s M/s** (STR phase %.0f / %.0f / %.0f ms; round 5: 57.1 / 45.6 / 39.6; round 4: 56.4 / 40.8 / 34.5)" % (
                 M(imp["value"]), imp["roofline"]["phase_ms"]["hs_str_kernel"], M(inh[0]["value"]), M(inh[1]["value"]), M(inh[2]["value"]),
                 inh[0]["roofline"]["phase_ms"]["hs_str_kernel"], inh[1]["roofline"]["phase_ms"]["hs_str_kernel"], inh[2]["roofline"]["phase_ms"]["hs_str_kernel"])))
fl = {}
cur = None
for l in open(os.path.join(D, "r06_flow_rates.txt")):
    if l.startswith("=="): cur = l.strip("= \n")
    elif l.startswith("{"):
        d = json.loads(l); fl.setdefault((cur.split(" ")[0], d["loci"], d["threads"], d["stream"]), d["loci_per_s"])
g = lambda lib, t, s: fl.get((lib, 768, t, s), float("nan"))
rows.append(("**the reference's own `SeqStutterGenotyper::genotype()` on the MI355X, caller unedited** (`r06_flow_rates.txt`, 768 loci, identical digest `8b8fd681…` in every row incl. the CPU's)",
             "%.0f loci/s on one host thread, %.0f with 4 (%.0f sharing a stream), **%.0f with 16 and %.0f sharing a stream**; with the optional retrace body %.0f / %.0f / %.0f; CPU reference on 16 cores %.0f" % (
                 g("libflow_mi355x.so", 1, 0), g("libflow_mi355x.so", 4, 0), g("libflow_mi355x.so", 4, 1), g("libflow_mi355x.so", 16, 0), g("libflow_mi355x.so", 16, 1),
                 g("libflow_mi355x_batched.so", 1, 0), g("libflow_mi355x_batched.so", 16, 0), g("libflow_mi355x_batched.so", 16, 1), g("libflow_ref.so", 16, 0))))
pl = ns["pipeline"]
rows.append(("the other device stages (`pipeline` block of the NS line)", "Needleman-Wunsch %.2f M pairs/s; traceback %.0f k/s (32 loci per call); stutter EM %.1f k loci/s (512 loci); the python-orchestrated chain on 64 loci %.2f k loci/s" % (
    pl["needleman_wunsch"]["pairs_per_s"] / 1e6, pl["traceback"]["tracebacks_per_s"] / 1e3, pl["stutter_em"]["loci_per_s"] / 1e3, pl["chain"]["loci_per_s"] / 1e3)))
cb = ns["cpu_baseline"]
rows.append(("**CPU baseline**: compiled reference on the GPU box's host, same generator (`cpu_baseline` block)", "one core %.1f k alignments/s; 16 processes (the cgroup's allowance) %.1f k alignments/s" % (cb["single_core"]["value"] / 1e3, cb["value"] / 1e3)))
print("| | value |\n|---|---|")
for a, b in rows: print("| %s | %s |" % (a, b))
