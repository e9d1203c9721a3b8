"""WAV writer of the output side (afg_wav_encode, host only; reference wav.d:365-701): header layout and the
reference's sample conversions, read back with the standard library."""
import io
import struct
import wave

import numpy as np
import pytest

import afgpu
import oraclelib


def header(data):
    riff, riff_len, wav, fmt, fmt_len, tag, ch, rate, bps, align, bits, dat, dat_len = struct.unpack("<4sI4s4sIHHIIHH4sI", data[:44])
    return dict(riff=riff, riff_len=riff_len, wave=wav, fmt=fmt, fmt_len=fmt_len, tag=tag, channels=ch, rate=rate,
                bytes_per_sec=bps, align=align, bits=bits, data=dat, data_len=dat_len)


@pytest.mark.parametrize("fmt,ss,tag", [(afgpu.WAV_S8, 1, 1), (afgpu.WAV_S16LE, 2, 1), (afgpu.WAV_S24LE, 3, 1),
                                        (afgpu.WAV_FP32LE, 4, 3), (afgpu.WAV_FP64LE, 8, 3)])
def test_header_fields(fmt, ss, tag):
    x = np.linspace(-1, 1, 7 * 3, dtype=np.float32).reshape(7, 3)
    d = afgpu.wav_encode(x, 22050, fmt)
    h = header(d)
    assert (h["riff"], h["wave"], h["fmt"], h["data"]) == (b"RIFF", b"WAVE", b"fmt ", b"data")
    assert h["fmt_len"] == 16 and h["tag"] == tag and h["channels"] == 3 and h["rate"] == 22050
    assert h["align"] == 3 * ss and h["bits"] == 8 * ss and h["bytes_per_sec"] == 22050 * 3 * ss
    assert h["data_len"] == 7 * 3 * ss and h["riff_len"] == 4 + 24 + 8 + h["data_len"]      # wav.d:573
    assert len(d) == 44 + h["data_len"]                                                       # no pad byte


def test_pcm_conversions_are_the_reference_formulas():
    rng = np.random.default_rng(2)
    x = np.concatenate([np.array([-1.0, 1.0, 0.0, 1e-6, -1e-6, 0.5, -0.5], np.float32),
                        rng.uniform(-1, 1, 500).astype(np.float32)])[:, None]
    xd = x[:, 0].astype(np.float64)
    d16 = afgpu.wav_encode(x, 8000, afgpu.WAV_S16LE)
    assert np.array_equal(np.frombuffer(d16[44:], "<i2"), ((32768.5 + xd * 32767.0).astype(np.int64) - 32768).astype(np.int16))
    d24 = np.frombuffer(afgpu.wav_encode(x, 8000, afgpu.WAV_S24LE)[44:], np.uint8).reshape(-1, 3).astype(np.int32)
    v24 = d24[:, 0] | (d24[:, 1] << 8) | (d24[:, 2] << 16)
    v24 = np.where(v24 & 0x800000, v24 - (1 << 24), v24)
    assert np.array_equal(v24, (8388608.5 + xd * 8388607.0).astype(np.int64) - 8388608)
    d8 = np.frombuffer(afgpu.wav_encode(x, 8000, afgpu.WAV_S8)[44:], np.uint8)
    assert np.array_equal(d8, (128.5 + xd * 127.0).astype(np.int64).astype(np.uint8))
    assert np.array_equal(np.frombuffer(afgpu.wav_encode(x, 8000, afgpu.WAV_FP32LE)[44:], "<f4"), x[:, 0])
    assert np.array_equal(np.frombuffer(afgpu.wav_encode(x, 8000, afgpu.WAV_FP64LE)[44:], "<f8"), xd)


def test_python_wave_module_reads_the_pcm_file():
    x = (0.25 * np.sin(np.arange(1000) * 0.05)).astype(np.float32)
    st = np.stack([x, -x], 1)
    w = wave.open(io.BytesIO(afgpu.wav_encode(st, 44100, afgpu.WAV_S16LE)))
    assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (2, 2, 44100, 1000)
    got = np.frombuffer(w.readframes(1000), "<i2").reshape(-1, 2)
    assert np.abs(got[:, 0] / 32767.0 - x).max() < 1 / 32767 and np.array_equal(got[:, 0], -got[:, 1])


def test_bad_arguments():
    with pytest.raises(afgpu.AfgError):
        afgpu.wav_encode(np.zeros((4, 2), np.float32), 44100, 9)
    assert afgpu.lib().afg_wav_encoded_size(10, 2000, afgpu.WAV_S16LE) == 0              # > 1024 channels, wav.d:400
    assert afgpu.wav_encode(np.zeros((0, 2), np.float32), 44100)[40:44] == b"\0\0\0\0"   # empty file: header only


def _lcg(seed):
    state = [seed & 0x7fffffff]

    def draw():
        state[0] = (state[0] * 1103515245 + 12345) & 0x7fffffff
        return state[0]
    return draw


def _pcm_of(data, fmt):
    raw = np.frombuffer(data[44:], np.uint8)
    if fmt == afgpu.WAV_S8:
        return raw.view(np.int8).astype(np.int32)
    if fmt == afgpu.WAV_S16LE:
        return raw.view("<i2").astype(np.int32)
    b = raw.reshape(-1, 3).astype(np.int32)
    v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
    return np.where(v & 0x800000, v - (1 << 24), v)


@pytest.mark.parametrize("fmt,bits", [(afgpu.WAV_S8, 8), (afgpu.WAV_S16LE, 16), (afgpu.WAV_S24LE, 24)])
def test_tpdf_dither_matches_the_restatement_on_the_same_draws(fmt, bits):
    """TPDFDither.process (wav.d:674-701) in front of the integer conversions (wav.d:483, :497, :513), fed the same
    generator as the oracle's restatement: identical bytes; with dither off the old behaviour."""
    rng = np.random.default_rng(bits)
    x = np.concatenate([np.array([-1.0, 1.0, 0.0, 0.999999, -0.999999, 1e-7], np.float32),
                        rng.uniform(-1, 1, 3000).astype(np.float32), (0.3 * np.sin(np.arange(500) * 0.01)).astype(np.float32)])
    got = _pcm_of(afgpu.wav_encode(x[:, None], 44100, fmt, dither=_lcg(7)), fmt)
    want = oraclelib.wav_pcm(x, bits, dither=_lcg(7))
    assert np.array_equal(got, want)
    assert np.array_equal(_pcm_of(afgpu.wav_encode(x[:, None], 44100, fmt), fmt), oraclelib.wav_pcm(x, bits))        # dither off
    plain = oraclelib.wav_pcm(x, bits)
    assert (got != plain).any() and np.abs(got - plain).max() <= 1                   # dither moves samples by at most one step
    # two draws per sample, in order: a different seed gives a different file
    assert (_pcm_of(afgpu.wav_encode(x[:, None], 44100, fmt, dither=_lcg(8)), fmt) != got).any()


def test_tpdf_dither_statistics():
    """What the dither is for: a constant between two 8-bit steps comes out as a mix of both.  The reference's tuning adds
    0.3125 + U1/4 + U2/8 before the floor (wav.d:688-692), i.e. a triangular-ish offset in [0.3125, 0.6875] of mean 0.5:
    a level half-way between two steps lands on either with probability 1/2, one within 0.31 of a step always on it."""
    def s8(level, dither):
        x = np.full(20000, level / 127.0, np.float32)
        raw = np.frombuffer(afgpu.wav_encode(x[:, None], 8000, afgpu.WAV_S8, dither=dither)[44:], np.uint8)
        return raw.astype(np.int32) - 128                     # 8-bit WAV is offset binary (cast(byte)(128.5 + x*127), wav.d:487)
    half = s8(40.5, _lcg(3))
    assert set(np.unique(half)) == {40, 41} and abs(float((half == 41).mean()) - 0.5) < 0.03
    assert set(np.unique(s8(40.6, None))) == {41} and set(np.unique(s8(40.4, None))) == {40}    # without dither: plain rounding
    assert set(np.unique(s8(40.2, _lcg(3)))) == {40} and set(np.unique(s8(40.8, _lcg(3)))) == {41}


def test_default_generator_is_libc_rand():
    """rng = NULL draws from libc rand() like the reference: reseeding libc reproduces the file, and the oracle
    restatement drawing from the same libc stream agrees."""
    import ctypes
    libc = ctypes.CDLL(None)
    x = np.random.default_rng(0).uniform(-1, 1, 2000).astype(np.float32)
    libc.srand(1234)
    a = afgpu.wav_encode(x[:, None], 44100, afgpu.WAV_S16LE, dither="libc")
    libc.srand(1234)
    b = afgpu.wav_encode(x[:, None], 44100, afgpu.WAV_S16LE, dither="libc")
    libc.srand(1234)
    want = oraclelib.wav_pcm(x, 16, dither="libc")
    assert a == b and np.array_equal(_pcm_of(a, afgpu.WAV_S16LE), want)
