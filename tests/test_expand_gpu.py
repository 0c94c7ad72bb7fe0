"""GPU: the tables the device builds for itself (hipstr_amd/csrc/expand_kernels.hip) against the host code they replace.

Round 4 moved the constants and closed-form tables of fully periodic STR options, and the 256-byte per-allele records of
hs_str_group_kernel_p, from the host preparation to two kernels queued behind the upload.  HIPSTR_HOST_TABLES=1 keeps the host code
(prep.cpp emit_stropt) for the constants and tables: every option's 20 constants and its table must come out bit for bit the same
either way, every record must be the same up to where its table sits, and so must the forward scores."""
import ctypes as C
import os

import numpy as np
import pytest

from hipstr_amd import capi

pytestmark = pytest.mark.gpu

STROPT_INTS = 40          # sizeof(hs_stropt_t) / 4 (hipstr_amd/csrc/layout.h)
F_B, F_PERIOD, F_F64, F_SHAPE, F_TABOFF, F_TABLEN, F_TABBASE, F_NDEQ, F_TAIL, F_KIND, F_GEN = 1, 3, 4, 19, 26, 27, 28, 35, 36, 37, 38


def _tables(hmm, sb, host_tables):
    if host_tables:
        os.environ["HIPSTR_HOST_TABLES"] = "1"
    else:
        os.environ.pop("HIPSTR_HOST_TABLES", None)
    try:
        dev = hmm.hipstr_hmm_upload(sb.ptr)
        assert dev, hmm.hipstr_last_error()
    finally:
        os.environ.pop("HIPSTR_HOST_TABLES", None)
    out = []
    for what, dt in ((0, np.int32), (1, np.float64), (2, np.int32)):
        n = hmm.hipstr_debug_fetch_table(dev, what, None, 0)
        assert n >= 0, hmm.hipstr_last_error()
        a = np.zeros(max(1, n // np.dtype(dt).itemsize), dt)
        assert hmm.hipstr_debug_fetch_table(dev, what, a.ctypes.data_as(C.c_void_p), n) == n
        out.append(a[:n // np.dtype(dt).itemsize])
    assert hmm.hipstr_hmm_align(dev, None) == 0
    p = np.zeros(sb.n_out); s = np.zeros(sb.n_reads, np.int32)
    assert hmm.hipstr_hmm_fetch(dev, p.ctypes.data_as(capi._f64p), s.ctypes.data_as(capi._i32p)) == 0
    hmm.hipstr_hmm_free(dev)
    return out[0].reshape(-1, STROPT_INTS), out[1], out[2].reshape(-1, 64), p, s


@pytest.mark.parametrize("kw,env", [
    (dict(n_loci=40, reads_per_locus=30, n_str_alleles=12, seed=5), {}),
    (dict(n_loci=30, reads_per_locus=20, n_str_alleles=8, read_len=150, flank_len=35, str_bp=40, seed=6), {}),
    (dict(n_loci=25, reads_per_locus=25, n_str_alleles=10, n_flank_opts=2, seed=7, mask_rate=0.2), {"HIPSTR_SYNTH_IMPERFECT": "0.5"}),
    (dict(n_loci=20, reads_per_locus=17, n_str_alleles=7, read_len=40, flank_len=12, str_bp=8, seed=21), {}),
    (dict(n_loci=4, reads_per_locus=40, n_str_alleles=20, read_len=300, flank_len=140, str_bp=300, seed=31), {}),
], ids=["c2like", "p30like", "flanks-imperfect-masks", "tiny", "long"])
def test_device_tables_equal_host_tables(hmm, kw, env):
    os.environ.update(env)
    try:
        sb = capi.SynthBatch(**kw)
    finally:
        for k in env:
            os.environ.pop(k, None)
    so_d, f64_d, rec_d, p_d, s_d = _tables(hmm, sb, False)
    so_h, f64_h, rec_h, p_h, s_h = _tables(hmm, sb, True)
    assert so_d.shape == so_h.shape and rec_d.shape == rec_h.shape
    assert np.all(so_d[:, F_GEN] == 0) and np.all(so_h[:, F_GEN] == 0)           # every generated option was finished
    n_gen = 0
    same = [F_B, F_PERIOD, F_TABLEN, F_NDEQ, F_TAIL, F_KIND] + list(range(F_SHAPE, F_SHAPE + 7)) + list(range(F_TABBASE, F_TABBASE + 7))
    for a, b in zip(so_d, so_h):
        assert np.array_equal(a[same], b[same])
        ca, cb = f64_d[a[F_F64]:a[F_F64] + 20], f64_h[b[F_F64]:b[F_F64] + 20]
        assert ca.tobytes() == cb.tobytes()                                          # stutter pmf | priors
        if a[F_TABLEN] > 0:
            n = 3 * a[F_TABLEN] + 1
            assert f64_d[a[F_TABOFF]:a[F_TABOFF] + n].tobytes() == f64_h[b[F_TABOFF]:b[F_TABOFF] + n].tobytes()
        n_gen += int(a[F_F64] != b[F_F64])
    assert n_gen > len(so_d) // 2                                                     # most options took the device path
    keep = [j for j in range(64) if j != 3]
    assert np.array_equal(rec_d[:, keep], rec_h[:, keep])
    for ra, rb in zip(rec_d, rec_h):                                                  # [3] = where the record's table sits: same content there
        n = 3 * ((ra[0] >> 10) & 0xff) + 1
        assert f64_d[ra[3]:ra[3] + n].tobytes() == f64_h[rb[3]:rb[3] + n].tobytes()
    assert np.array_equal(s_d, s_h) and p_d.tobytes() == p_h.tobytes()
