"""GPU: cr_math.h on the device returns what the same header returns on the host, bit for bit — 4 M arguments per function over the
ranges test_cr_math.py covers (the host side is pinned there against decimal, glibc and long double), plus the special values.  The
device has its own fma / rint / ldexp and subnormal handling: this is where a difference would show."""
import ctypes as C

import numpy as np
import pytest

from hipstr_amd import capi
import test_cr_math as host

pytestmark = pytest.mark.gpu
_f64p = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def lib():
    return host.build_lib()


def _dev(hmm, which, x):
    x = np.ascontiguousarray(x, np.float64); y = np.empty_like(x)
    assert hmm.hipstr_debug_cr_math(which, x.ctypes.data_as(_f64p), y.ctypes.data_as(_f64p), x.size) == 0, hmm.hipstr_last_error()
    return y


def test_exp_device_equals_host(hmm, lib):
    rng = np.random.default_rng(5)
    pw = 2.0 ** -np.arange(20, 70).astype(np.float64)
    x = np.concatenate([rng.uniform(-746, 710, 1_500_000), rng.uniform(-40, 0, 1_500_000), rng.uniform(-1, 1, 500_000), rng.uniform(-745.2, -707, 300_000),
                        rng.uniform(-2.0 ** -27, 2.0 ** -27, 200_000), pw, -pw, np.nextafter(pw, 1), -np.nextafter(pw, 0),
                        [0.0, -0.0, 709.782712893384, 709.7827128933841, 710.0, -745.13, -745.14, -745.2, -800.0, np.inf, -np.inf, 5e-324, -5e-324, 1e-300]])
    got = _dev(hmm, 0, x); want = host._batch(lib.cr_exp_batch, x)
    bad = np.nonzero(got.view(np.uint64) != want.view(np.uint64))[0]
    assert bad.size == 0, [(x[i].hex(), got[i].hex(), want[i].hex()) for i in bad[:10]]
    assert np.isnan(_dev(hmm, 0, [np.nan]))[0]


def test_log_device_equals_host(hmm, lib):
    rng = np.random.default_rng(6)
    x = np.concatenate([rng.uniform(1, 2 ** 24, 1_500_000), rng.uniform(0.5, 2, 1_000_000), 1 + rng.uniform(-2.0 ** -6, 2.0 ** -6, 500_000), np.exp(rng.uniform(-744, 709, 800_000)),
                        rng.uniform(0, 4e-308, 200_000), [1.0, np.nextafter(1.0, 2), np.nextafter(1.0, 0), 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, np.inf]])
    x = x[x > 0]
    got = _dev(hmm, 1, x); want = host._batch(lib.cr_log_batch, x)
    bad = np.nonzero(got.view(np.uint64) != want.view(np.uint64))[0]
    assert bad.size == 0, [(x[i].hex(), got[i].hex(), want[i].hex()) for i in bad[:10]]
    r = _dev(hmm, 1, [0.0, -1.0, np.nan])
    assert r[0] == -np.inf and np.isnan(r[1]) and np.isnan(r[2])
