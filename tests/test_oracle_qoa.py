"""Pins the QOA oracle (oracle/qoa_lms.c): decode(encode(pcm)) must equal the encoder's own
reconstruction sample for sample (the format's defining property), on top of hand-checked LMS steps."""
import numpy as np
import pytest

import oraclelib
import afgpu


def tone(n, ch, seed=0, amp=9000):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    x = np.stack([amp * np.sin(2 * np.pi * (0.01 + 0.003 * c) * t + c) + 300 * rng.standard_normal(n) for c in range(ch)], 1)
    return np.clip(np.round(x), -32768, 32767).astype(np.int16)


@pytest.mark.parametrize("n,ch", [(5120 * 2 + 777, 2), (5120, 1), (19, 2), (12345, 3)])
def test_decode_reproduces_encoder_reconstruction(n, ch):
    pcm = tone(n, ch, seed=n)
    data, recon = oraclelib.qoa_encode(pcm)
    frames, channels, rate, total = afgpu.qoa_frames(data.tobytes())
    assert channels == ch and rate == 44100 and total == n and int(frames["samples"].sum()) == n
    oi, of = oraclelib.qoa_transform(frames, data, n * ch)
    assert (oi.reshape(n, ch) == recon).all()
    assert (of == oi.astype(np.float32) * np.float32(1.0 / 32767)).all()           # qoa.d:831-838
    err = oi.reshape(n, ch).astype(np.int64) - pcm
    if n >= 5000:
        assert np.sqrt(np.mean(err.astype(np.float64) ** 2)) < 400                  # lossy but close (~ -30 dB)


def test_first_samples_by_hand():
    # silence encodes to quantised residuals whose dequantised value is +-1 (scalefactor 0): with the
    # start weights {0,0,-8192,16384} and zero history the first prediction is 0 (qoa.d:231-239, :578-582)
    data, recon = oraclelib.qoa_encode(np.zeros((40, 1), np.int16))
    frames, *_ = afgpu.qoa_frames(data.tobytes())
    oi = oraclelib.qoa_transform(frames, data, 40, want_float=False)
    assert (oi == recon.reshape(-1)).all() and np.abs(oi).max() <= 1
    hdr = int.from_bytes(bytes(data[8:16]), "big")
    assert (hdr >> 56) == 1 and ((hdr >> 16) & 0xffff) == 40 and (hdr & 0xffff) == 8 + 16 + 8 * 2
    w = int.from_bytes(bytes(data[24:32]), "big")
    assert [(w >> s) & 0xffff for s in (48, 32, 16, 0)] == [0, 0, 0xE000, 0x4000]
