"""Builds every native artefact in-tree (so the .so files travel with the repo snapshot).

  hipstr_amd/csrc/libhipstr_hmm.so    product: HIP kernels (gfx950) + C-ABI          [hipcc]
  hipstr_amd/synth/libhipstr_synth.so bench/test input generator                     [g++]
  oracle/libhipstr_oracle.so          TEST INFRASTRUCTURE: C restatement              [gcc, oracle/Makefile]
  oracle/_ref/libhipstr_ref.so        TEST INFRASTRUCTURE: the compiled reference     [only where /root/reference exists]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hipstr_amd", "csrc")
HIP_SOURCES = ["api.hip", "hmm_kernels.hip", "expand_kernels.hip", "post_kernels.hip", "prep.cpp", "trace.hip", "em.hip", "nw.hip", "batch_io.cpp", "stream.hip", "gather.cpp"]
HIP_HEADERS = ["exports.map", "layout.h", "post_layout.h", "prep.h", "device_common.h", "api_internal.h", os.path.join("..", "..", "include", "hipstr_hmm.h"), os.path.join("..", "..", "include", "hipstr_hmm_debug.h")]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-honor-nans", "-mno-amdgpu-ieee", "-fPIC", "-shared", "-pthread", "-fvisibility=hidden", "-Wno-unused-result", "-Wno-unused-value"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _run(cmd, cwd=None):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=cwd)


def build_hmm(force=False):
    """One object per source (in parallel, under build/hmm_obj: git-ignored, rebuilt where a source or a header is newer), then the link —
    the same flags and result as one hipcc call over all sources, in the time of the longest file instead of their sum."""
    out = os.path.join(CSRC, "libhipstr_hmm.so")
    hdrs = [os.path.join(CSRC, s) for s in HIP_HEADERS]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = ["-D%s=%s" % (m, os.environ[e]) for m, e in (("HS_TRAIL_ROWS", "HIPSTR_TRAIL_ROWS"), ("HS_STR_WAVES", "HIPSTR_STR_WAVES")) if os.environ.get(e)]
    objdir = os.path.join(ROOT, "build", "hmm_obj" + ("_" + "_".join(extra).replace("-D", "").replace("=", "") if extra else ""))
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"]
    jobs, objs = [], []
    for s in HIP_SOURCES:
        o = os.path.join(objdir, os.path.splitext(s)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [os.path.join(CSRC, s)] + hdrs):
            cmd = [hipcc] + cflags + extra + ["-c", "-o", o + ".tmp", os.path.join(CSRC, s)]
            print("+", " ".join(cmd), flush=True)
            jobs.append((subprocess.Popen(cmd), o, cmd))
    failed = []
    for pr, o, cmd in jobs:
        if pr.wait() != 0:
            failed.append(" ".join(cmd))
        else:
            os.replace(o + ".tmp", o)
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    if force or jobs or _stale(out, objs + [os.path.join(CSRC, "exports.map")]):
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), "-o", out] + objs)
    return out


def build_synth(force=False):
    src = os.path.join(ROOT, "hipstr_amd", "synth", "synth.cpp")
    out = os.path.join(ROOT, "hipstr_amd", "synth", "libhipstr_synth.so")
    if force or _stale(out, [src, os.path.join(ROOT, "include", "hipstr_hmm.h")]):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", out, src])
    return out


def build_oracle(force=False):
    _run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"] + (["-B"] if force else []))
    if os.path.isdir("/root/reference/src"):
        _run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref", "-j8"])


def build_dropin():
    """integration/HapAlignerMI355X (the reference-side binding) + its equivalence check, only where the HipSTR tree exists."""
    if os.path.isdir("/root/reference/src"):
        _run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "dropin"])
        # the reference's own SeqStutterGenotyper::genotype() on top of the adapter (integration/genotype_flow.cpp)
        _run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "flow", "-j8"])


def build_all(force=False):
    build_synth(force)
    build_oracle(force)
    build_hmm(force)
    build_dropin()


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
