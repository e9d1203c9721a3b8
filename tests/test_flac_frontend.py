"""Host front-ends of the outer surface (no device): afg_flac_parse against an independent FLAC writer,
afg_qoa_parse against the frame headers, and the error-state contract of the stream API.

Reference behaviour followed: drflac.d:1444-1569 (headers), :1279-1328 (residual), :1901-2118 (metadata),
qoa.d:413-486, stream.d:295-316 (error state), internals.d:16-23 (messages)."""
import numpy as np
import pytest

import afgpu
import flac_bitstream as fb
import flac_ref_encoder as enc
import oraclelib


def make_pcm(n, channels, bps, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    amp = (1 << (bps - 1)) * 0.4
    cols = []
    for c in range(channels):
        x = amp * np.sin(0.01 * (c + 1) * t + c) + amp * 0.05 * rng.standard_normal(n)
        cols.append(np.clip(np.round(x), -(1 << (bps - 1)), (1 << (bps - 1)) - 1))
    return np.stack(cols, 1).astype(np.int64)


def check_records(parsed, want):
    info, frames, subframes, res = parsed
    wf, ws, wr, total = want
    assert info["out_samples"] == total
    for k in ("in_off", "out_off", "block_size", "sf_index", "channels", "assignment", "bps"):
        np.testing.assert_array_equal(frames[k], wf[k], err_msg=k)
    for k in ("coef", "order", "shift", "wasted", "use64"):
        np.testing.assert_array_equal(subframes[k], ws[k], err_msg=k)
    np.testing.assert_array_equal(res, wr)


@pytest.mark.parametrize("channels,bps,block,n,rate", [
    (2, 16, 4096, 4096 * 3 + 777, 44100),     # the C4 shape + a ragged last block (explicit 16-bit size)
    (2, 24, 1152, 1152 * 2 + 100, 96000),     # 64-bit predictor path, explicit 8-bit size
    (1, 8, 576, 576 * 3, 11025),              # mono, sample rate code 13
    (2, 12, 192, 192 * 4 + 1, 50000),         # one-sample last block, sample rate code 12
    (3, 20, 256, 256 * 3, 655350),            # independent multichannel, sample rate code 14
])
def test_parse_recovers_encoder_records(channels, bps, block, n, rate):
    pcm = make_pcm(n, channels, bps, 7 + channels)
    data, want = fb.encode_file(pcm, bps, block, sample_rate=rate, orders=(8, 12, 3, 32) if block > 64 else (2,))
    parsed = afgpu.flac_parse(data)
    info = parsed[0]
    assert (info["sample_rate"], info["channels"], info["bps"]) == (rate, channels, bps)
    assert info["total_samples"] == n and info["max_block"] == block
    check_records(parsed, want)
    # host front-end + the oracle's restore stage == the PCM that was encoded
    out = oraclelib.flac_transform(parsed[1], parsed[2], parsed[3], info["out_samples"])
    got = out.reshape(-1, channels) >> (32 - bps)
    keep = np.ones(n, bool)
    for fr in parsed[1]:
        # drflac.d:2894-2919 applies a subframe's wasted-bits shift to the OUTPUT channel, after the stereo
        # decorrelation, so a decorrelated frame with wasted bits does not reproduce its input in the reference
        # (nor here, by design); such frames (a side channel that happens to be all-even) are left out.
        if fr["assignment"] >= 8 and parsed[2]["wasted"][fr["sf_index"]:fr["sf_index"] + 2].any():
            lo = int(fr["out_off"]) // channels
            keep[lo:lo + int(fr["block_size"])] = False
    assert keep.sum() >= n - block
    np.testing.assert_array_equal(got[keep], pcm[keep])


def test_wasted_bits_constant_and_verbatim_subframes():
    n, block = 1024, 256
    pcm = make_pcm(n, 2, 16, 3)
    pcm[:, 1] = (pcm[:, 1] >> 3) << 3            # 3 wasted bits on the right channel
    pcm[256:512, 0] = 1234                        # a constant block
    pcm[512:768, 0] = np.random.default_rng(5).integers(-30000, 30000, 256)   # noise: still decodes
    data, want = fb.encode_file(pcm, 16, block, assignments=(enc.INDEPENDENT,), use_fixed_every=2, orders=(0, 1))
    parsed = afgpu.flac_parse(data)
    check_records(parsed, want)
    assert parsed[2]["wasted"].max() == 3
    out = oraclelib.flac_transform(parsed[1], parsed[2], parsed[3], parsed[0]["out_samples"])
    np.testing.assert_array_equal(out.reshape(-1, 2) >> 16, pcm)


def test_variable_blocksize_stream_rice2_and_metadata():
    pcm = make_pcm(4096 + 512, 2, 16, 11)
    frames, subframes, res, total = enc.encode(pcm, 16, 4096)
    meta = [(1, bytes(100)), (2, b"afgp" + bytes(8)), (4, bytes(40))]          # PADDING, APPLICATION, VORBIS_COMMENT
    data = fb.write_file(frames, subframes, res, 44100, 16, extra_metadata=meta, variable=True, rice2=True,
                         header_bps=False)
    check_records(afgpu.flac_parse(data), (frames, subframes, res, total))


def test_truncation_and_trailing_garbage_stop_at_the_last_good_frame():
    pcm = make_pcm(1024 * 4, 2, 16, 2)
    data, want = fb.encode_file(pcm, 16, 1024)
    whole = afgpu.flac_parse(data)
    assert len(whole[1]) == 4
    cut = afgpu.flac_parse(data[:len(data) - 40])                 # last frame loses its tail
    assert len(cut[1]) == 3 and cut[0]["out_samples"] == 3 * 1024 * 2
    np.testing.assert_array_equal(cut[3], whole[3][:3 * 2048])
    junk = afgpu.flac_parse(data + bytes(range(1, 200)))
    assert len(junk[1]) == 4
    only_header = afgpu.flac_parse(data[:4 + 4 + 34])
    assert len(only_header[1]) == 0 and only_header[0]["total_samples"] == 4096


def test_reserved_codes_are_rejected():
    pcm = make_pcm(512, 1, 16, 1)
    frames, subframes, res, total = enc.encode(pcm, 16, 256, orders=(4,), use_fixed_every=1000)
    good = fb.write_file(frames, subframes, res, 44100, 16)
    assert len(afgpu.flac_parse(good)[1]) == 2
    first = 4 + 4 + 34
    bad = bytearray(good)
    bad[first + 3] |= 0x06                                         # bits-per-sample code 3: reserved
    assert len(afgpu.flac_parse(bytes(bad))[1]) == 0
    bad = bytearray(good)
    bad[first + 2] = (bad[first + 2] & 0x0f)                       # block size code 0: reserved
    assert len(afgpu.flac_parse(bytes(bad))[1]) == 0
    bad = bytearray(good)
    bad[first] = 0x7f                                              # broken sync
    assert len(afgpu.flac_parse(bytes(bad))[1]) == 0


def test_escape_partition_follows_the_reference():
    """drflac.d:1301/:1304 compare the Rice parameter with 16/32, so an escaped partition is read as Rice
    parameter 15: the frame does not come out as the encoder meant.  Identical-to-reference means the same here:
    the stream must not decode to the original samples (and must not crash)."""
    pcm = make_pcm(256, 1, 16, 9)
    frames, subframes, res, total = enc.encode(pcm, 16, 256, orders=(2,), use_fixed_every=1000)
    data = fb.write_file(frames, subframes, res, 44100, 16, escape_partition=0, escape_bits=16)
    info, fr, sf, rs = afgpu.flac_parse(data)
    assert len(fr) == 0 or not np.array_equal(rs, res)


def test_not_flac():
    for blob in (b"", b"fLa", b"RIFF" + bytes(100), b"fLaC" + bytes(3), b"fLaC\x80\x00\x00\x10" + bytes(16)):
        with pytest.raises(afgpu.AfgError):
            afgpu.flac_parse(blob)


def test_qoa_parse_matches_the_headers():
    n = 5120 * 2 + 333
    t = np.arange(n)
    pcm = np.stack([8000 * np.sin(0.02 * t), 6000 * np.sin(0.013 * t)], 1).round().astype(np.int16)
    data, _ = oraclelib.qoa_encode(pcm, 22050)
    frames, ch, sr, smp = afgpu.qoa_parse(data.tobytes())
    want, wch, wsr, wsmp = afgpu.qoa_frames(data.tobytes())
    assert (ch, sr, smp) == (wch, wsr, wsmp) == (2, 22050, n)
    for k in ("byte_off", "out_off", "samples", "channels"):
        np.testing.assert_array_equal(frames[k], want[k])
    assert list(frames["samples"]) == [5120, 5120, 333]
    cut, *_ = afgpu.qoa_parse(data.tobytes()[:-9])                  # truncated last frame is dropped
    assert len(cut) == 2
    with pytest.raises(afgpu.AfgError):
        afgpu.qoa_parse(b"qoaf" + bytes(4) + bytes(16))              # zero samples: qoa.d:438
    with pytest.raises(afgpu.AfgError):
        afgpu.qoa_parse(b"qoax" + bytes(40))


def test_stream_error_state_contract():
    """stream.d:31-33: nothing throws; a failed open leaves the stream in error state with a message."""
    s = afgpu.AudioStream()
    assert s.isError() and s.errorMessage() == "Stream not initialized"           # stream.d:1379
    s.openFromMemory(b"definitely not audio" * 10)
    assert s.isError()
    assert s.errorMessage() == "Cannot decode stream: unrecognized encoding."     # internals.d:16
    assert s.getFormat() == afgpu.FORMAT_UNKNOWN and s.getNumChannels() == 0
    assert s.getLengthInFrames() == afgpu.UNKNOWN_LENGTH
    assert s.readSamplesFloat(np.zeros(8, np.float32)) == 0
    s.cleanUp()


def test_stream_without_device_fails_loudly():
    """No CPU fallback: a recognised file on a box without a gfx950 device is an error, never host-decoded."""
    if afgpu.device_count() > 0:
        pytest.skip("a device is present")
    data, _ = fb.encode_file(make_pcm(512, 2, 16, 1), 16, 256)
    s = afgpu.AudioStream()
    s.openFromMemory(data)
    assert s.isError() and s.errorMessage() == "Decoder initialization failed"    # internals.d:18
    with pytest.raises(afgpu.AfgError):
        afgpu.batch_decode([data])


@pytest.mark.parametrize("case", ["spikes", "silence", "noise24", "noise24_rice2", "short_tail"])
def test_rice_decoder_extremes(case):
    """the residual loop's windowed fast path (host/afg_flac_front.cpp) against the writer's records where it has to hand over
    to the bit reader: unary runs far longer than a 64-bit window (rare spikes under a Rice parameter chosen for silence),
    k = 0 throughout, 5-bit parameters with 20+ low bits, and symbols in the last eight bytes of the file"""
    rng = np.random.default_rng(11)
    kw = {}
    if case == "spikes":
        n, bps = 4096 * 2, 16
        pcm = np.zeros((n, 2), np.int64)
        at = rng.integers(20, n - 20, 24)
        pcm[at, rng.integers(0, 2, 24)] = rng.integers(-30000, 30000, 24)
    elif case == "silence":
        n, bps = 4096 + 9, 16
        pcm = np.zeros((n, 2), np.int64)
        pcm[:, 1] = 1
    elif case.startswith("noise24"):
        n, bps = 1152 * 2 + 5, 24
        pcm = rng.integers(-(1 << 23), 1 << 23, (n, 2))
        kw["rice2"] = case.endswith("rice2")
    else:
        n, bps = 4096 + 3, 16
        pcm = make_pcm(n, 2, 16, 5)
    data, want = fb.encode_file(pcm, bps, 4096 if bps == 16 else 1152, orders=(8, 2), **kw)
    parsed = afgpu.flac_parse(data)
    check_records(parsed, want)
    # ... and the same file with nothing behind its last frame byte but also with padding behind it: the fast path's
    # "eight whole bytes left" test must not change a value
    parsed2 = afgpu.flac_parse(data + bytes(16))
    np.testing.assert_array_equal(parsed2[3][:len(parsed[3])], parsed[3])
