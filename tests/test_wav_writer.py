"""WAV writer of the output side (afg_wav_encode, host only; reference wav.d:365-701): header layout and the
reference's sample conversions, read back with the standard library."""
import io
import struct
import wave

import numpy as np
import pytest

import afgpu


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
