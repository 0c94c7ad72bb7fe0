"""CPU: hipstr_amd/csrc/cr_math.h (correctly rounded exp / log, used by the posterior, genotype and EM kernels so that they reproduce the
host libm's bits) — the same header compiled for the host:
  * against Python's decimal module (exp / ln correctly rounded at 60 digits, then rounded once to double): identical bits on 60 000
    arguments chosen where the implementation branches — table interval edges, the neighbourhood of 1 for log, tiny and huge arguments,
    subnormal results and arguments, the overflow / underflow edges;
  * against the host libm on 2 x 10^8 arguments (exp on [-746, 710] and on [-40, 0], the range the log-sum-exps use; log on
    [1, 2^24] and log-uniform over the whole double range): every disagreement is decided by long double (64-bit significand) — it must
    side with cr_math every time; arguments long double cannot decide go to decimal.  The disagreement rate is glibc's own
    misrounding rate (it documents < 1 ulp, not correct rounding)."""
import ctypes as C
import os
import struct
import subprocess
from decimal import Decimal, getcontext

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "cr_math_test.c")
LIB = os.path.join(ROOT, "tests", "cpp", "libcr_math_test.so")
HDRS = [os.path.join(ROOT, "hipstr_amd", "csrc", f) for f in ("cr_math.h", "cr_tables.inc")]
_f64p = C.POINTER(C.c_double)


def build_lib():
    if not os.path.exists(LIB) or any(os.path.getmtime(f) > os.path.getmtime(LIB) for f in [SRC] + HDRS):
        flags = ["-mfma"] if "fma" in open("/proc/cpuinfo").read() else []        # without hardware fma the C library's (exact) fma is used: slower, same bits
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-fPIC", "-shared", "-Wall"] + flags + ["-o", LIB, SRC, "-lm"])
    l = C.CDLL(LIB)
    for f in ("cr_exp_batch", "cr_log_batch", "libm_exp_batch", "libm_log_batch"):
        getattr(l, f).argtypes = [_f64p, _f64p, C.c_int64]; getattr(l, f).restype = None
    l.cr_sweep_exp.argtypes = [C.c_uint64, C.c_int64, C.c_double, C.c_double, C.POINTER(C.c_int64), _f64p, C.c_int]; l.cr_sweep_exp.restype = None
    l.cr_sweep_log.argtypes = [C.c_uint64, C.c_int64, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_int64), _f64p, C.c_int]; l.cr_sweep_log.restype = None
    l.cr_quick_exp_check.argtypes = [C.c_uint64, C.c_int64, C.c_double, C.c_double, C.POINTER(C.c_int64)]; l.cr_quick_exp_check.restype = None
    return l


@pytest.fixture(scope="module")
def lib():
    return build_lib()


def _batch(fn, x):
    x = np.ascontiguousarray(x, np.float64); y = np.empty_like(x)
    fn(x.ctypes.data_as(_f64p), y.ctypes.data_as(_f64p), x.size)
    return y


def _bits(a):
    return np.asarray(a, np.float64).view(np.uint64)


def _dec_exp(x):
    getcontext().prec = 60
    d = Decimal(float(x)).exp()
    if d < Decimal(2) ** -1080:
        return 0.0
    try:
        return float(d)                 # one rounding, to nearest even, subnormals included
    except OverflowError:
        return float("inf")


def _dec_log(x):
    getcontext().prec = 60
    return float(Decimal(float(x)).ln())


def _from_bits(u):
    return struct.unpack("<d", struct.pack("<Q", int(u)))[0]


def test_exp_against_decimal(lib):
    rng = np.random.default_rng(1)
    ln2_128 = np.log(2.0) / 128
    xs = [rng.uniform(-745.2, 709.8, 8000), rng.uniform(-40, 0, 8000), rng.uniform(-1, 1, 4000), rng.uniform(-1e-3, 1e-3, 2000),
          rng.uniform(-745.2, -707.0, 4000),                                       # subnormal results
          np.array([0.0, -0.0, 1.0, -1.0, 709.782712893384, 709.7827128933841, -745.13, -745.14, -745.2, -708.3964185322641, -708.4, 2.0 ** -53, -2.0 ** -53, 2.0 ** -54,
                    2.0 ** -1022, 1e-300, -1e-300, 5e-324, 88.7, -87.3])]
    pw = 2.0 ** -np.arange(20, 70).astype(np.float64)                             # 1 + x an exact tie between doubles: only x^2/2 breaks it
    xs += [pw, -pw, np.nextafter(pw, 1), np.nextafter(pw, 0), -np.nextafter(pw, 1), -np.nextafter(pw, 0), 3 * pw, -3 * pw, rng.uniform(-2.0 ** -27, 2.0 ** -27, 3000),
           rng.uniform(-2.0 ** -52, 2.0 ** -52, 3000)]
    k = rng.integers(-130000, 130000, 4000)                                        # the reduction's interval edges: (k + 1/2) ln2/128 and its neighbours
    edges = (k + 0.5) * ln2_128
    xs += [edges, np.nextafter(edges, np.inf), np.nextafter(edges, -np.inf), k * ln2_128]
    x = np.concatenate(xs); x = x[(x <= 709.79) & (x >= -746)]
    got = _batch(lib.cr_exp_batch, x)
    want = np.array([_dec_exp(v) for v in x])
    bad = np.nonzero(_bits(got) != _bits(want))[0]
    assert bad.size == 0, [(x[i].hex(), got[i].hex(), want[i].hex()) for i in bad[:10]]
    assert np.isnan(_batch(lib.cr_exp_batch, [np.nan]))[0] and _batch(lib.cr_exp_batch, [710.0, np.inf, -np.inf, -800.0]).tolist() == [np.inf, np.inf, 0.0, 0.0]


def test_log_against_decimal(lib):
    rng = np.random.default_rng(2)
    xs = [rng.uniform(1, 2 ** 24, 8000), rng.uniform(1, 2, 6000), rng.uniform(0.5, 1, 4000), 1 + rng.uniform(-2 ** -6, 2 ** -6, 6000), 1 + rng.uniform(-1e-9, 1e-9, 3000),
          np.exp(rng.uniform(-744, 709, 8000)), rng.uniform(0, 1, 4000), rng.uniform(0, 4e-308, 2000),       # incl. subnormal arguments
          np.array([np.nextafter(1.0, 2), np.nextafter(1.0, 0), 2.0, 0.5, 4.0, 1e308, 1.7976931348623157e308, 5e-324, 2.2250738585072014e-308, 3.0, 10.0, np.e])]
    OFF = 0x3FE6A09E00000000
    edge = np.array([_from_bits(OFF + (i << 45)) for i in range(129)])               # table interval edges and their neighbours, in several binades
    for sc in (1.0, 2.0 ** 10, 2.0 ** -300):
        xs += [edge * sc, np.nextafter(edge, np.inf) * sc, np.nextafter(edge, -np.inf) * sc]
    x = np.concatenate(xs); x = x[x > 0]
    got = _batch(lib.cr_log_batch, x)
    want = np.array([_dec_log(v) for v in x])
    bad = np.nonzero(_bits(got) != _bits(want))[0]
    assert bad.size == 0, [(x[i].hex(), got[i].hex(), want[i].hex()) for i in bad[:10]]
    r = _batch(lib.cr_log_batch, [1.0, 0.0, np.inf])
    assert r[0] == 0.0 and r[1] == -np.inf and r[2] == np.inf and np.isnan(_batch(lib.cr_log_batch, [-1.0, np.nan])).all()


def _sweep(call, what, dec):
    counts = (C.c_int64 * 5)(); und = np.zeros(256)
    call(counts, und.ctypes.data_as(_f64p), 256)
    n, dis, mine, theirs, undecided = list(counts)
    assert theirs == 0, "%s: long double sides with libm against cr_math on %d arguments" % (what, theirs)
    # what long double cannot decide: decimal does
    for x in und[:min(undecided, 256)]:
        got = _batch(dec[0], [x])[0]
        assert _bits(got) == _bits(dec[1](x)), (what, float(x).hex())
    assert dis == mine + undecided
    return n, dis


def test_exp_against_libm_1e8(lib):
    tot = 0; dis = 0
    for seed, n, lo, hi in ((11, 50_000_000, -40.0, 0.0), (12, 50_000_000, -746.0, 710.0)):
        a, b = _sweep(lambda c, u, cap: lib.cr_sweep_exp(seed, n, lo, hi, c, u, cap), "exp [%g, %g]" % (lo, hi), (lib.cr_exp_batch, _dec_exp))
        tot += a; dis += b
    assert tot == 100_000_000
    print("exp: %d of %d arguments where the host libm is not correctly rounded (%.2e)" % (dis, tot, dis / tot))
    assert dis < tot // 100                       # glibc: well under 1 %


def test_log_against_libm_1e8(lib):
    tot = 0; dis = 0
    for seed, n, lo, hi, lu in ((21, 50_000_000, 1.0, 2.0 ** 24, 0), (22, 50_000_000, 1e-300, 1e300, 1)):
        a, b = _sweep(lambda c, u, cap: lib.cr_sweep_log(seed, n, lo, hi, lu, c, u, cap), "log [%g, %g]" % (lo, hi), (lib.cr_log_batch, _dec_log))
        tot += a; dis += b
    assert tot == 100_000_000
    print("log: %d of %d arguments where the host libm is not correctly rounded (%.2e)" % (dis, tot, dis / tot))
    assert dis < tot // 100


def test_exp_quick_phase_never_accepts_a_wrong_result(lib):
    """cr_exp's quick phase (plain double on a double-double reduced argument, accepted when its error bound 2^-64 cannot change the
    rounding) against the accurate double-double phase on 2 x 10^8 arguments: whatever it accepts is the accurate phase's result, and it
    accepts all but about one argument in a thousand."""
    tot = acc = 0
    for seed, n, lo, hi in ((31, 100_000_000, -40.0, 0.0), (32, 60_000_000, -708.0, 709.0), (33, 40_000_000, -1.0, 1.0)):
        c = (C.c_int64 * 3)()
        lib.cr_quick_exp_check(seed, n, lo, hi, c)
        assert c[2] == 0, "quick phase accepted %d wrong results on [%g, %g]" % (c[2], lo, hi)
        tot += c[0]; acc += c[1]
    print("exp quick phase: accepted %d of %d (%.4f %% go on to the accurate phase)" % (acc, tot, 100.0 * (tot - acc) / tot))
    assert acc > 0.995 * tot
