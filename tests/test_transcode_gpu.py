"""BASELINE config C1: the reference's `transcode` example (examples/transcode/source/main.d:12-84) end to end --
a real MP3 (and its Ogg twin) -> AudioStream surface -> 1024-frame chunks -> WAV -- run as a program, checked on
the decoded floats against the oracle front-end + transform, and on the dithered 24-bit bytes against the oracle's
restatement of WAVEncoder.writeSamples fed the same generator."""
import os
import subprocess
import sys

import numpy as np
import pytest

import afgpu
import oraclelib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def run_transcode(*args):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "transcode.py"), *args], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    return res.stdout


def expected_mp3():
    data = open(os.path.join(GOLDEN, "mathjax_invalid_keypress.mp3"), "rb").read()
    dec = oraclelib.mp3_decode_file(data)
    return dec["pcm"].reshape(-1, dec["channels"]), dec


def test_transcode_mp3_to_float_wav_matches_the_oracle_decode(gpu, tmp_path):
    want, dec = expected_mp3()
    out = tmp_path / "keypress.wav"
    log = run_transcode("--format", "f32", os.path.join(GOLDEN, "mathjax_invalid_keypress.mp3"), str(out))
    assert "format     = mp3" in log and f"samplerate = {dec['hz']} Hz" in log and f"channels   = {dec['channels']}" in log
    assert f"=> {len(want)} frames decoded" in log
    raw = out.read_bytes()
    assert raw[:4] == b"RIFF" and raw[8:12] == b"WAVE" and len(raw) == 44 + want.size * 4
    got = np.frombuffer(raw[44:], "<f4").reshape(-1, dec["channels"])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))                 # the decoded floats, bit for bit


def test_transcode_default_is_dithered_s24_like_the_example(gpu, tmp_path):
    """main.d:54-56: sampleFormat s24, enableDither true.  With a seeded generator the bytes are the restatement's."""
    import transcode
    want, dec = expected_mp3()
    out = tmp_path / "keypress24.wav"
    run_transcode("--dither-seed", "99", os.path.join(GOLDEN, "mathjax_invalid_keypress.mp3"), str(out))
    raw = out.read_bytes()
    assert len(raw) == 44 + want.size * 3 and raw[34:36] == (24).to_bytes(2, "little")
    b = np.frombuffer(raw[44:], np.uint8).reshape(-1, 3).astype(np.int32)
    got = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
    got = np.where(got & 0x800000, got - (1 << 24), got)
    assert np.array_equal(got, oraclelib.wav_pcm(want.reshape(-1), 24, dither=transcode.lcg(99)))
    # and without a seed the generator is libc rand(), as in the reference: the file differs from the undithered one
    plain = tmp_path / "plain.wav"
    dith = tmp_path / "dith.wav"
    run_transcode("--no-dither", os.path.join(GOLDEN, "mathjax_invalid_keypress.mp3"), str(plain))
    run_transcode(os.path.join(GOLDEN, "mathjax_invalid_keypress.mp3"), str(dith))
    assert plain.read_bytes()[:44] == dith.read_bytes()[:44] and plain.read_bytes() != dith.read_bytes()


def test_transcode_ogg_and_qoa_output(gpu, tmp_path):
    data = open(os.path.join(GOLDEN, "mathjax_invalid_keypress.ogg"), "rb").read()
    want = oraclelib.vorbis_file_pcm(oraclelib.vorbis_decode_file(data))
    out = tmp_path / "keypress_ogg.wav"
    log = run_transcode("--format", "f32", os.path.join(GOLDEN, "mathjax_invalid_keypress.ogg"), str(out))
    assert "format     = ogg" in log
    got = np.frombuffer(out.read_bytes()[44:], "<f4").reshape(-1, want.shape[1])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # QOA output (transcode's other target): decodes back through the library to the encoder's own reconstruction
    q = tmp_path / "keypress.qoa"
    run_transcode(os.path.join(GOLDEN, "mathjax_invalid_keypress.ogg"), str(q))
    back = afgpu.batch_decode([q.read_bytes()])[0]
    assert back["status"] == 0 and back["format"] == afgpu.FORMAT_QOA and back["frames"] == len(want)
    assert np.abs(back["pcm"] - np.clip(want, -1, 1)).max() < 0.05
