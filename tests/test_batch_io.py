"""Host only: the flat on-disk form of a batch (hipstr_batch_write / _read / _serialize / _deserialize) round-trips every array,
rejects damaged images, and a batch read back gives the oracle the same alignments."""
import ctypes as C
import os

import numpy as np
import pytest

from hipstr_amd import capi
import util


def _api(lib):
    lib.hipstr_batch_serialized_size.restype = C.c_int64; lib.hipstr_batch_serialized_size.argtypes = [capi._BP]
    lib.hipstr_batch_serialize.restype = C.c_int; lib.hipstr_batch_serialize.argtypes = [capi._BP, C.c_void_p, C.c_int64]
    lib.hipstr_batch_deserialize.restype = C.c_void_p; lib.hipstr_batch_deserialize.argtypes = [C.c_void_p, C.c_int64]
    lib.hipstr_batch_write.restype = C.c_int; lib.hipstr_batch_write.argtypes = [C.c_char_p, capi._BP]
    lib.hipstr_batch_read.restype = C.c_void_p; lib.hipstr_batch_read.argtypes = [C.c_char_p]
    lib.hipstr_batch_file_batch.restype = C.POINTER(capi.HipstrBatch); lib.hipstr_batch_file_batch.argtypes = [C.c_void_p]
    lib.hipstr_batch_file_free.restype = None; lib.hipstr_batch_file_free.argtypes = [C.c_void_p]
    return lib


@pytest.mark.parametrize("kw", [dict(n_loci=3, reads_per_locus=7, n_str_alleles=4, seed=1),
                                dict(n_loci=5, reads_per_locus=6, n_str_alleles=5, n_flank_opts=2, seed=2, mask_rate=0.3)])
def test_file_round_trip(hmm_host, oracle, tmp_path, kw):
    lib = _api(hmm_host)
    sb = capi.SynthBatch(**kw)
    path = str(tmp_path / "shard.hsb").encode()
    assert lib.hipstr_batch_write(path, sb.ptr) == 0
    assert os.path.getsize(path) == lib.hipstr_batch_serialized_size(sb.ptr)
    f = lib.hipstr_batch_read(path)
    assert f, lib.hipstr_last_error()
    back = lib.hipstr_batch_file_batch(f)
    a = util.batch_to_dict(util.synth_to_batch(sb)); b = util.batch_to_dict(util.synth_to_batch(type("V", (), {"ptr": back})()))
    assert a.keys() == b.keys() and all(np.array_equal(a[k], b[k]) for k in a)
    want = capi.run_align(oracle, "oracle_", sb.ptr, fill=-2.0); got = capi.run_align(oracle, "oracle_", back, fill=-2.0)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    lib.hipstr_batch_file_free(f)


def test_damaged_images_are_rejected(hmm_host):
    lib = _api(hmm_host)
    sb = capi.SynthBatch(n_loci=2, reads_per_locus=4, n_str_alleles=3, seed=3)
    n = lib.hipstr_batch_serialized_size(sb.ptr)
    buf = (C.c_uint8 * n)()
    assert lib.hipstr_batch_serialize(sb.ptr, buf, n) == 0
    assert lib.hipstr_batch_serialize(sb.ptr, buf, n - 1) != 0
    good = bytes(buf)
    f = lib.hipstr_batch_deserialize(good, n); assert f; lib.hipstr_batch_file_free(f)
    for bad, why in ((good[:-8], b"truncated"), (b"X" + good[1:], b"magic"), (good[:200] + bytes([good[200] ^ 1]) + good[201:], b"checksum")):
        assert not lib.hipstr_batch_deserialize(bad, len(bad)) and why in lib.hipstr_last_error()


def _refresh_checksum(img):
    h = 1469598103934665603
    for byte in img[64:]:
        h = ((h ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return img[:32] + h.to_bytes(8, "little") + img[40:]


def test_forged_images_with_valid_checksum_are_rejected(hmm_host):
    """The checksum is unkeyed: an image whose arrays contradict each other but whose checksum was recomputed must still be
    refused before any of its entries is used as an index (sizes come from the section table only)."""
    lib = _api(hmm_host)
    sb = capi.SynthBatch(n_loci=2, reads_per_locus=4, n_str_alleles=3, seed=3)
    n = lib.hipstr_batch_serialized_size(sb.ptr)
    buf = (C.c_uint8 * n)()
    assert lib.hipstr_batch_serialize(sb.ptr, buf, n) == 0
    good = bytes(buf)
    elem = [4, 4, 4, 4, 8, 4, 1, 4, 1, 4, 4, 1, 1, 4, 4, 1, 4, 1]
    counts = [int.from_bytes(good[64 + 16 * s + 8:64 + 16 * s + 16], "little") for s in range(18)]
    start, q = [], 64 + 16 * 18
    for s in range(18):
        start.append(q); q += (elem[s] * counts[s] + 7) & ~7

    def poke(sec, index, value):
        at = start[sec] + 4 * index
        return _refresh_checksum(good[:at] + int(value).to_bytes(4, "little", signed=True) + good[at + 4:])

    forged = {
        "huge option count": poke(2, 1, 1 << 28),          # blk_nopts -> would index far past opt_off
        "zero option count": poke(2, 0, 0),
        "hap_off end": poke(7, 2, 1 << 20),                # hap_off[n] -> realign_hap size
        "read_off decreases": poke(9, 1, -5),
        "read_off end": poke(9, 2, 1 << 20),               # read_off[n] -> would index past base_off
        "base_off non-monotone": poke(10, 2, 1),
        "base_off negative start": poke(10, 0, -1),
        "cigar_off end": poke(14, counts[14] - 1, 7),
        "opt_off end": poke(5, counts[5] - 1, 1 << 24),
    }
    assert lib.hipstr_batch_deserialize(_refresh_checksum(good), n)       # the helper itself produces acceptable images
    for name, img in forged.items():
        assert not lib.hipstr_batch_deserialize(img, len(img)), name
        assert b"do not match" in lib.hipstr_last_error(), name


def test_empty_batch(hmm_host):
    lib = _api(hmm_host)
    b = capi.Batch().finalize()
    n = lib.hipstr_batch_serialized_size(b.ptr)
    buf = (C.c_uint8 * n)()
    assert lib.hipstr_batch_serialize(b.ptr, buf, n) == 0
    f = lib.hipstr_batch_deserialize(bytes(buf), n)
    assert f and lib.hipstr_batch_file_batch(f).contents.n_loci == 0
    lib.hipstr_batch_file_free(f)
