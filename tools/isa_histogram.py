#!/usr/bin/env python
"""Static instruction-class histogram of every kernel of hmm_kernels.hip / post_kernels.hip / expand_kernels.hip / em.hip, from the compiler's gfx950 assembly
(hipcc --cuda-device-only -S with the library's flags; no GPU needed) -> profiles/<tag>_isa_histogram.json.

Classes follow the measured issue costs (profiles/*_valu_microbench.json):
  fp64      v_add/max/min/mul/fma_f64                                   4 cycles per wave64 instruction
  wide      anything that writes SGPR/VCC (v_cmp*), crosses lanes (v_readlane/v_writelane/readfirstlane/DPP/SDWA), converts from/to f64,
            or exists only in the VOP3 encoding / is emitted as _e64 (v_cndmask_b32_e64, v_lshl_add_u32, v_mad_*, v_mul_lo_u32, ...)   4 cycles
  simple32  the remaining 32-bit VOP1/VOP2 (_e32) instructions: v_mov_b32, v_add_u32, v_and_b32, v_cndmask_b32_e32 ...              2 cycles
tools/sq_counters.py uses (a) max/min per add among the FP64 instructions — the SQ_INSTS_VALU_ADD_F64 counter does not see
v_max_f64 — and (b) the wide : simple32 split of the rest, to turn instruction counters into VALU pipe time.  Static counts
weight every basic block once, so (b) is an approximation; the brackets low/high in the counter summary do not depend on it.
usage: python tools/isa_histogram.py > profiles/<tag>_isa_histogram.json"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hipstr_amd import build   # noqa: E402

VOP3_ONLY = re.compile(r"^v_(lshl_add|add3|lshl_or|and_or|or3|xad|mad_|fma_f32|bfe_|bfi_|perm_|alignbit|alignbyte|mul_lo|mul_hi|lshlrev_b64|lshrrev_b64|ashrrev_i64|"
                       r"lshl_add_u64|add_lshl|med3|min3|max3|div_|ldexp_f64|frexp|trig|cvt_pk|mbcnt|sad_|cubeid|pk_)")


def classify(m):
    if not m.startswith("v_"):
        return None
    if re.match(r"^v_(add|max|min|mul|fma)_f64", m):
        return "fp64"
    if m.startswith(("v_cmp", "v_readlane", "v_writelane", "v_readfirstlane", "v_permlane", "v_swap")) or "_dpp" in m or "_sdwa" in m:
        return "wide"
    if re.match(r"^v_cvt_.*f64", m) or m.endswith("_e64") or VOP3_ONLY.match(m) or re.search(r"_(f64|b64|u64|i64)(_e32)?$", m):
        return "wide"
    return "simple32"


out = {}
for src in ("hmm_kernels.hip", "post_kernels.hip", "expand_kernels.hip", "em.hip"):
    with tempfile.TemporaryDirectory() as t:
        asm = os.path.join(t, "k.s")
        flags = [f for f in build.HIPCC_FLAGS if f not in ("-shared", "-fPIC", "-pthread")]
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["--cuda-device-only", "-S", os.path.join(build.CSRC, src), "-o", asm],
                              stderr=subprocess.DEVNULL)
        text = open(asm).read()
    for name in re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", text, re.M):
        m = re.search(r"^%s:.*?\n(.*?)\n\.Lfunc_end\d+:" % re.escape(name), text, re.S | re.M)
        if not m:
            continue
        body = m.group(1)
        # the key is the name rocprofv3 reports, without namespaces and arguments: hs_trail_kernel_coop<15, 4, 3> (tools/sq_counters.py key_of)
        short = subprocess.check_output(["c++filt", name]).decode().strip().split("::")[-1].split("(")[0]
        mn = collections.Counter()
        cls = collections.Counter()
        for line in body.split("\n"):
            s = line.strip()
            if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
                continue
            op = s.split()[0]
            c = classify(op)
            if c:
                mn[op] += 1; cls[c] += 1
        if sum(cls.values()) == 0:
            continue
        f_add = sum(v for k, v in mn.items() if re.match(r"^v_(add|mul|fma)_f64", k)); f_mm = sum(v for k, v in mn.items() if re.match(r"^v_(max|min)_f64", k))
        rest = cls["wide"] + cls["simple32"]
        out[short] = {"symbol": name, "valu_static": sum(cls.values()), "classes": dict(cls),
                      "fp64_maxmin_per_addmulfma": (f_mm / f_add) if f_add else 0.0,
                      "wide_share_of_non_fp64": (cls["wide"] / rest) if rest else 0.0,
                      "top_mnemonics": dict(mn.most_common(12))}
print(json.dumps({"flags": " ".join(build.HIPCC_FLAGS), "kernels": out}, indent=1))
