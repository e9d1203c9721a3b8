"""Pins the FLAC oracle (oracle/flac_restore.c): lossless encode -> decode round trips against
an independent encoder model (tests/flac_ref_encoder.py), plus the int32/int64 accumulator
semantics of drflac.d:1060-1140."""
import ctypes as C

import numpy as np
import pytest

import flac_ref_encoder as enc
import oraclelib

L = oraclelib.lib()


def music_like(rng, n, channels, bps):
    t = np.arange(n)
    base = sum(rng.uniform(0.05, 0.3) * np.sin(2 * np.pi * rng.uniform(0.001, 0.08) * t + rng.uniform(0, 6)) for _ in range(5))
    x = np.stack([base + 0.02 * rng.standard_normal(n) + 0.1 * c * np.sin(0.01 * t) for c in range(channels)], 1)
    amp = (1 << (bps - 1)) - 1
    return np.clip(np.round(x / np.abs(x).max() * amp * 0.9), -amp, amp).astype(np.int64)


@pytest.mark.parametrize("bps,block,channels", [(16, 4096, 2), (16, 1152, 2), (24, 4096, 2), (16, 577, 1), (8, 192, 2)])
def test_round_trip_is_lossless(bps, block, channels):
    rng = np.random.default_rng(bps * 1000 + block)
    pcm = music_like(rng, block * 5 + 123, channels, bps)
    frames, subframes, res, total = enc.encode(pcm, bps, block)
    out = oraclelib.flac_transform(frames, subframes, res, total)
    want = (pcm.reshape(-1) << (32 - bps)).astype(np.int64)
    assert (out.astype(np.int64) == want).all()
    assert {int(a) for a in frames["assignment"]} == ({0, 8, 9, 10} if channels == 2 else {0})


def test_wasted_bits_round_trip():
    rng = np.random.default_rng(4)
    pcm = music_like(rng, 4096, 2, 16) & ~3            # two wasted bits in every channel
    frames, subframes, res, total = enc.encode(pcm, 16, 1024, assignments=(enc.INDEPENDENT,))
    assert subframes["wasted"].min() >= 2
    out = oraclelib.flac_transform(frames, subframes, res, total)
    assert (out.astype(np.int64) == (pcm.reshape(-1) << 16)).all()


def test_float_conversion_matches_stream_d():
    rng = np.random.default_rng(9)
    pcm = music_like(rng, 2000, 2, 16)
    frames, subframes, res, total = enc.encode(pcm, 16, 1000)
    out, outf = oraclelib.flac_transform(frames, subframes, res, total, want_float=True)
    want = (out.astype(np.float64) * (1.0 / 2147483647.0)).astype(np.float32)      # stream.d:507-510
    assert (outf.view(np.uint32) == want.view(np.uint32)).all()


def test_prediction_accumulators():
    coef = np.array([32767, -32768, 12345] + [0] * 29, np.int16)
    hist = np.array([2 ** 31 - 1, -2 ** 31, 77777, 0], np.int32)      # p[-3], p[-2], p[-1], p[0]
    p = hist.ctypes.data + 3 * 4
    exact = 32767 * 77777 + (-32768) * (-2 ** 31) + 12345 * (2 ** 31 - 1)
    for shift in (0, 5, 14, 31):
        w32 = ((exact + 2 ** 31) % 2 ** 32) - 2 ** 31
        assert L.afgo_flac_prediction_32(3, shift, coef.ctypes.data, p) == w32 >> shift
        w64 = (((exact >> shift) + 2 ** 31) % 2 ** 32) - 2 ** 31
        assert L.afgo_flac_prediction_64(3, shift, coef.ctypes.data, p) == w64
    assert L.afgo_flac_prediction_32(0, 3, coef.ctypes.data, p) == 0


def test_int16_residual_rows_are_the_same_samples():
    """the int16 storage of residual rows (include/afg.h AFG_FLAC_ROW16) is a storage format only"""
    from afgpu import synthetic
    frames, subframes, res, total = synthetic.flac_batch(5, n_frames=40, vary_block=True, orders=(0, 3, 8, 12))
    want = oraclelib.flac_transform(frames, subframes, res, total)
    pf, pres = synthetic.flac_pack16(frames, res, 2)
    assert pf["res16"].sum() >= 15 and (pf["in_off"][pf["res16"] == 1] % 8 == 0).all()
    assert (oraclelib.flac_transform(pf, subframes, pres, total) == want).all()
