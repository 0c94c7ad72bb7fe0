#!/usr/bin/env python
"""Per-kernel SQ counters from two rocprofv3 --pmc passes (own runs, kernel trace only; tools/profile_pass.sh) -> <tag>_sq_counters.json.

What binds these kernels is VALU issue, and instructions do not all cost the same: on gfx950 a wave64 FP64 add/max/fma occupies
its SIMD for 4 cycles, a plain 32-bit VOP1/VOP2 (mov, add, and) for 2, and everything that writes an SGPR/VCC, crosses lanes or
needs the VOP3 encoding (v_cmp, v_readlane, DPP, v_cndmask with an SGPR mask, v_cvt_*_f64) for 4 — measured, not assumed
(tools/valu_microbench.hip -> profiles/*_valu_microbench.json).  The counters give the split: FP64 arithmetic instructions =
(ADD_F64 + MUL_F64 + FMA_F64) x (1 + max,min per add,mul,fma) — the counters do not see v_max_f64 / v_min_f64 (the flank kernels
execute 8 adds and 4 maxima per cell and ADD_F64 reads 44 % of SQ_INSTS_VALU, not 67 %), so their share comes from the kernel's
static instruction histogram (tools/isa_histogram.py) — and the rest.  For the rest the pipe time is bracketed and estimated:
    pipe_occupancy_low   every non-FP64 VALU instruction at 2 cycles
    pipe_occupancy_high  every non-FP64 VALU instruction at 4 cycles  (what round 1 assumed for ALL instructions)
    pipe_occupancy_est   non-FP64 instructions split into 4-cycle and 2-cycle classes in the static histogram's proportion
SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE are kept as collected (SQ_* in quad-cycles summed over SIMDs or SEs).
fp64_frac_of_peak = FP64 instructions x 64 lanes / (duration x 256 CU x 4 SIMD x 16 lanes x clock): the FP64-VALU roofline.
usage: python tools/sq_counters.py <sq1.db> <sq2.db> <alignments_per_launch> "<command>" <profiles dir> > out.json"""
import glob
import hashlib
import json
import os
import re
import sqlite3
import sys

CLOCK, SIMDS = 2.4e9, 1024
db1, db2, n_aln, cmd, prof_dir = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4], sys.argv[5]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def key_of(name):
    return name.split("::")[-1].split("(")[0]


def passes_of(command):
    """Passes of the hot path in a profiled run: bench.py --steps S --warmup W runs S + W of them."""
    m1, m2 = re.search(r"--steps\s+(\d+)", command), re.search(r"--warmup\s+(\d+)", command)
    return max(1, (int(m1.group(1)) if m1 else 1) + (int(m2.group(1)) if m2 else 0))


def collect(path):
    """Counters and duration of every kernel PER PASS: the average over its dispatches times its dispatches per pass (a pass that takes its
    reads in several chunks launches a kernel once per chunk: bench.py scales these figures by alignments per pass)."""
    db = sqlite3.connect(path)
    per_pass = {}
    for name, n in db.execute("select kernel_name, count(distinct dispatch_id) from counters_collection group by kernel_name"):
        per_pass[name] = n / float(passes_of(cmd))
    kern = {}
    for name, counter, avg in db.execute("select kernel_name,counter_name,avg(value) from counters_collection group by kernel_name,counter_name"):
        if "hs_" in name:
            kern.setdefault(key_of(name), {})[counter] = avg * per_pass[name]
    dur = {}
    for name, d in db.execute("select kernel_name, avg(duration) from (select distinct dispatch_id, kernel_name, duration from counters_collection) group by kernel_name"):
        dur[key_of(name)] = d * per_pass[name]
    return kern, dur


k1, d1 = collect(db1)
k2, d2 = collect(db2)
micro = sorted(glob.glob(os.path.join(prof_dir, "r*_valu_microbench.json")))
cyc = {"fp64": 4.0, "simple32": 2.0, "other": 4.0, "source": None}
if micro:
    m = json.load(open(micro[-1]))["cycles_per_wave64_instruction"]
    # the microbenchmark loop adds ~7 % of loop overhead at 4 wavefronts per SIMD; the issue cost is the rounded value
    cyc = {"fp64": float(round(m["v_add_f64"]["4_waves_per_simd"])), "simple32": float(round(m["v_mov_b32"]["4_waves_per_simd"])),
           "other": float(round(m["v_cmp_eq_u32"]["4_waves_per_simd"])), "source": os.path.basename(micro[-1])}
hist = {}
hfiles = sorted(glob.glob(os.path.join(prof_dir, "r*_isa_histogram.json")))
if hfiles:
    hist = json.load(open(hfiles[-1]))["kernels"]
out = {}
for k in sorted(k1):
    c = dict(k1[k]); c.update(k2.get(k, {}))
    dur = d1.get(k)
    if not dur:
        continue
    valu = c.get("SQ_INSTS_VALU", 0.0)
    h = hist.get(k)
    if h is None:
        # without the static histogram the FP64 maxima are not counted and every other instruction is priced at 4 cycles: refuse instead
        sys.exit("tools/sq_counters.py: no static instruction histogram for kernel %r in %s (keys: %s) — regenerate it with tools/isa_histogram.py"
                 % (k, hfiles[-1] if hfiles else "profiles/", ", ".join(sorted(hist))))
    f64_counted = c.get("SQ_INSTS_VALU_ADD_F64", 0.0) + c.get("SQ_INSTS_VALU_MUL_F64", 0.0) + c.get("SQ_INSTS_VALU_FMA_F64", 0.0)
    f64 = f64_counted * (1.0 + h.get("fp64_maxmin_per_addmulfma", 0.0))
    rest = max(valu - f64, 0.0)
    wide = h.get("wide_share_of_non_fp64", 1.0)
    denom = SIMDS * CLOCK * dur * 1e-9
    out[k] = {"duration_ns": dur, "duration_ns_second_pass": d2.get(k), "counters": {n: c[n] for n in sorted(c)},
              "valu_insts": valu, "fp64_arith_insts_counted": f64_counted, "fp64_arith_insts": f64, "other_valu_insts": rest,
              "static_histogram": {"fp64_maxmin_per_addmulfma": h.get("fp64_maxmin_per_addmulfma"), "wide_share_of_non_fp64": h.get("wide_share_of_non_fp64")},
              "pipe_occupancy_est": (f64 * cyc["fp64"] + rest * (wide * cyc["other"] + (1.0 - wide) * cyc["simple32"])) / denom,
              "pipe_occupancy_low": (f64 * cyc["fp64"] + rest * cyc["simple32"]) / denom,
              "pipe_occupancy_high": (f64 * cyc["fp64"] + rest * cyc["other"]) / denom,
              "fp64_frac_of_peak": f64 * cyc["fp64"] / denom,
              "valu_insts_per_alignment": valu / n_aln}
src = os.path.join(ROOT, "hipstr_amd", "csrc", "hmm_kernels.hip")
print(json.dumps({"isa_histogram": os.path.basename(hfiles[-1]) if hfiles else None, "command": cmd, "alignments_per_launch": n_aln, "clock_hz": CLOCK, "simds": SIMDS, "cycles_per_wave64_instruction": cyc,
                  "kernel_source_sha1": hashlib.sha1(open(src, "rb").read()).hexdigest() if os.path.exists(src) else None,
                  "kernels": out}, indent=1))
