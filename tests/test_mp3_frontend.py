"""MP3 Layer III host front-end (no device): the product parser (afg_mp3_parse) against the oracle restatement of
minimp3 / minimp3_ex on a real file and on damaged variants of it, plus sanity of what the oracle decodes.

Reference behaviour followed: minimp3.d:487-1000 (side info .. reorder), :1170-1230 (reservoir, L3_decode),
:1436-1581 (sync, frame decode); minimp3_ex.d:93-190 (ID3/APE, Xing/Info), :566-639 (open), :787-888 (read)."""
import os

import numpy as np
import pytest

import afgpu
import oraclelib

FIXTURE = os.path.join(os.path.dirname(__file__), "golden", "mathjax_invalid_keypress.mp3")


def real():
    return open(FIXTURE, "rb").read()


def same_records(data):
    want = oraclelib.mp3_decode_file(data)
    if want is None:
        with pytest.raises(afgpu.AfgError):
            afgpu.mp3_parse(data)
        return None, None
    info, runs, coef, flags, copies = afgpu.mp3_parse(data)
    assert (info["channels"], info["hz"], info["tagged"], info["start_delay"]) == \
           (want["channels"], want["hz"], want["tagged"], want["start_delay"])
    assert info["detected_samples"] == want["detected_samples"] and info["declared_samples"] == want["declared_samples"]
    np.testing.assert_array_equal(runs, want["runs"])
    np.testing.assert_array_equal(flags & 0x00ffffff, want["flags"])                  # the reference's per-granule fields
    assert coef.shape == want["coef"].shape
    assert np.array_equal(coef.view(np.uint32), want["coef"].view(np.uint32))          # bit-exact spectra
    # AFG_MP3_NZ_BANDS (bits 24..29): exactly the subbands up to the last line that is not +0.0
    bits = coef.view(np.uint32).reshape(-1, 576)
    last = np.where(bits.any(1), 575 - np.argmax(bits[:, ::-1] != 0, 1), -1)
    np.testing.assert_array_equal((flags >> 24) & 63, (last + 18) // 18 + 1)
    assert info["pcm_samples"] == len(want["pcm"]) == int(copies[:, 1].sum())
    return (info, runs, coef, flags, copies), want


def delivered(parsed, pcm_plane):
    """apply the copy plan to a PCM plane (what the host does after the device stage)"""
    copies = parsed[4]
    return np.concatenate([pcm_plane[int(s):int(s + n)] for s, n in copies]) if len(copies) else np.zeros(0, np.float32)


def oracle_plane(parsed):
    info, runs, coef, flags, _ = parsed
    ch = info["channels"]
    return oraclelib.mp3_transform(runs, np.full(len(runs), ch, np.uint8), coef.reshape(-1), flags)


def test_real_file_records_and_delivery():
    parsed, want = same_records(real())
    info = parsed[0]
    assert (info["channels"], info["hz"], info["tagged"]) == (2, 44100, 1)
    assert info["start_delay"] == 2 * (576 + 529)                  # LAME delay 576 + the decoder's 529, both channels
    assert len(set(int(f) & 3 for f in parsed[3])) == 4             # normal, start, short and stop blocks all occur
    # records -> transform oracle -> copy plan == what the oracle's own frame-by-frame drive delivered
    got = delivered(parsed, oracle_plane(parsed))
    assert np.array_equal(got.view(np.uint32), want["pcm"].view(np.uint32))


def test_real_file_decodes_to_plausible_audio():
    """No second MP3 decoder exists in this image; what can be checked is that the waveform is continuous across
    frame boundaries (a reservoir / overlap / Huffman slip shows as a click), bounded and tonal."""
    want = oraclelib.mp3_decode_file(real())
    pcm = want["pcm"].reshape(-1, 2)
    assert len(pcm) == 23087 and np.isfinite(pcm).all() and 0.3 < np.abs(pcm).max() < 1.0
    x = pcm[:, 0].astype(np.float64)
    jumps = np.abs(np.diff(x))
    assert jumps.max() < 0.05 * np.abs(x).max()                   # the earcon is a low thud: no sample-to-sample jumps
    curv = np.abs(x[2:] - 2 * x[1:-1] + x[:-2])                   # second difference: a seam would stand out here
    edges = np.arange(1152 - 1105, len(x) - 3, 576)               # granule boundaries in delivered time
    loud = np.abs(x[1:-1]) > 0.05
    seam = max(curv[edges - 1].max(), curv[edges].max(), curv[edges + 1].max())
    assert seam <= np.percentile(curv[loud], 99) and curv.max() < 2e-3
    spec = np.abs(np.fft.rfft(x * np.hanning(len(x))))
    assert 80 < spec.argmax() * 44100 / len(x) < 400
    assert np.corrcoef(pcm[:, 0], pcm[:, 1])[0, 1] > 0.9            # near-mono content through M/S stereo


def test_tags_garbage_and_truncation():
    d = real()
    body = d[45:]                                                   # without the ID3v2 tag
    for variant in (body,
                    d + b"TAG" + bytes(125),                        # ID3v1
                    d + b"APETAGEX" + (2000).to_bytes(4, "little") + (32).to_bytes(4, "little") + bytes(16),
                    bytes(300) + body,                              # leading junk: sync search
                    b"\xff\xfb\x90" + body,                         # a false sync in front
                    d[:5000],                                       # cut inside a frame
                    d[:45 + 209],                                   # only the Info frame
                    body[209:]):                                    # no Info tag: full scan, reservoir start-up
        same_records(variant)
    parsed, _ = same_records(body[209:])
    assert parsed[0]["tagged"] == 0 and parsed[0]["start_delay"] == 0 and parsed[0]["detected_samples"] == 0


def test_damaged_frames_agree_with_the_oracle():
    d = bytearray(real())
    rng = np.random.default_rng(7)
    hits = 0
    for trial in range(60):
        v = bytearray(d)
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(45 + 209, len(v) - 2400))
            kind = int(rng.integers(0, 3))
            if kind == 0:
                v[pos] ^= 1 << int(rng.integers(0, 8))              # bit flip anywhere (side info, main data, header)
            elif kind == 1:
                del v[pos:pos + int(rng.integers(1, 300))]          # lost bytes: resynchronisation, new state run
            else:
                v[pos:pos] = bytes(rng.integers(0, 256, int(rng.integers(1, 50)), dtype=np.uint8))
        parsed, want = same_records(bytes(v))
        if parsed is not None and len(parsed[1]) > 1:
            hits += 1
    assert hits > 0                                                 # some variants did split into several state runs


def test_not_mp3():
    for blob in (b"", bytes(5000), bytes(range(1, 255)) * 20):
        with pytest.raises(afgpu.AfgError):
            afgpu.mp3_parse(blob)
    s = afgpu.AudioStream()
    s.openFromMemory(b"RIFF" + real())                              # containers the reference probes before MP3
    assert s.isError() and s.errorMessage() == "Cannot decode stream: unrecognized encoding."
