"""The K-level piecewise closed form of interrupted repeats (layout.h HS_SHAPE_PWK; prep.cpp piecewise_k; hmm_kernels.hip pwk_eval_grp) against the
entry-by-entry replay of the same visiting lists (StutterAlignerClass.cpp:59-150), on the host: the library's own preparation code builds the lists
and the descriptor slots of random interrupted blocks, tests/cpp/pwk_form_test.cpp evaluates both for every bound — bit for bit.  (The device
kernel is compared with the oracle in tests/test_hmm_gpu.py's interrupted-repeat cases.)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_k_level_form_equals_the_replay(tmp_path):
    exe = str(tmp_path / "pwk_form_test")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "pwk_form_test.cpp"), os.path.join(ROOT, "hipstr_amd", "csrc", "prep.cpp"),
                           "-lpthread", "-o", exe])
    out = subprocess.run([exe, "1500", "20260929"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    sys.stderr.write(out.stderr)
    assert out.returncode == 0, out.stdout
    head = out.stdout.splitlines()[0].split()
    kv = dict(zip(head[0::2], map(int, head[1::2])))
    assert kv["mismatches"] == 0
    assert kv["pwk"] > 1000 and kv["evaluations"] > 200000
    breaks = list(map(int, out.stdout.splitlines()[1].split()[1:]))
    assert all(b > 0 for b in breaks[3:7])          # three to six breaks all seen
