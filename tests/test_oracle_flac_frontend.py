"""The oracle's FLAC front-end (oracle/flac_frontend.c: the reference's 32-bit cache bit reader, frame / subframe parse
and fused Rice + prediction loop, drflac.d:680-1043, :1143-1328, :1444-1695, :2846-2960) and QOA stream layer
(oracle/qoa_lms.c: qoa.d:413-534, :803-851):

  * pinned by files of KNOWN content: an independent FLAC writer (tests/flac_bitstream.py) and the QOA encoder's own
    reconstruction;
  * then used as the expectation of the product's host parsers (afgpu.flac_parse / afgpu.qoa_frames, different code:
    a 64-bit window reader) on the same bytes, clean and damaged -- bit for bit, with the one class of input named
    below where the product stops a frame early on purpose.

No device: the product side here is host code + the oracle's restore stage."""
import os

import numpy as np
import pytest

import afgpu
import flac_bitstream as fb
import flac_ref_encoder as enc
import oraclelib
from test_flac_frontend import make_pcm

CONFIGS = [
    (2, 16, 4096, 4096 * 3 + 777, 44100),
    (2, 24, 1152, 1152 * 2 + 100, 96000),
    (1, 8, 576, 576 * 3, 11025),
    (2, 12, 192, 192 * 4 + 1, 50000),
    (3, 20, 256, 256 * 3, 655350),
]


def encoded(cfg, seed=0):
    channels, bps, block, n, rate = cfg
    pcm = make_pcm(n, channels, bps, 7 + channels + seed)
    data, want = fb.encode_file(pcm, bps, block, sample_rate=rate, orders=(8, 12, 3, 32) if block > 64 else (2,))
    return bytes(data), pcm, want


def product_pcm(data):
    """afg_flac_parse (product host code) + the oracle's restore stage; None if the product refuses the file"""
    try:
        info, frames, sub, res = afgpu.flac_parse(data)
    except Exception:
        return None
    if info["out_samples"] == 0:
        return np.zeros(0, np.int32)
    return oraclelib.flac_transform(frames, sub, res, info["out_samples"])


@pytest.mark.parametrize("cfg", CONFIGS)
def test_oracle_front_end_recovers_the_encoded_pcm(cfg):
    channels, bps, block, n, rate = cfg
    data, pcm, want = encoded(cfg)
    o = oraclelib.flac_decode_file(data)
    assert (o["channels"], o["sample_rate"], o["bps"], o["max_block"]) == (channels, rate, bps, block)
    assert o["total_samples"] == n * channels and o["flags"] == 0 and o["n_frames"] == len(want[0])
    got = o["pcm"].reshape(-1, channels) >> (32 - bps)
    keep = np.ones(n, bool)
    for fr in want[0]:
        # a decorrelated frame with wasted bits does not reproduce its input in the reference (drflac.d:2894-2919 shifts
        # the OUTPUT channel): left out, as in tests/test_flac_frontend.py
        if fr["assignment"] >= 8 and want[1]["wasted"][fr["sf_index"]:fr["sf_index"] + 2].any():
            lo = int(fr["out_off"]) // channels
            keep[lo:lo + int(fr["block_size"])] = False
    assert keep.sum() >= n - block
    np.testing.assert_array_equal(got[keep], pcm[keep])
    # and the product's parser says the same, every sample
    np.testing.assert_array_equal(product_pcm(data), o["pcm"])


def test_escape_codes_are_rice_parameters_in_both():
    """drflac.d:1301, :1304 test the Rice parameter against 16 / 32, which a 4- / 5-bit field never holds: a partition a FLAC
    encoder writes raw is decoded as Rice codes with parameter 15 / 31 by the reference, so by both parsers here -- the
    same (wrong) samples, or the same end of the stream."""
    pcm = make_pcm(512, 1, 16, 9)
    frames, subframes, res, total = enc.encode(pcm, 16, 256, orders=(2,), use_fixed_every=1000)
    for part in (0, 1):
        data = bytes(fb.write_file(frames, subframes, res, 44100, 16, escape_partition=part, escape_bits=16))
        o = oraclelib.flac_decode_file(data)
        p = product_pcm(data)
        want = o["pcm"] if not o["flags"] else o["pcm"][:o["first_flag_sample"]]
        np.testing.assert_array_equal(p, want)
        assert not np.array_equal(o["pcm"].reshape(-1, 1) >> 16, pcm[:len(o["pcm"])])


def test_last_frame_without_its_crc():
    """drflac__seek_bits hands whole bytes it cannot find to the client's seek (drflac.d:812-819), and AudioStream's seek
    callback reports success at any offset (stream.d:2227-2239): the last frame is delivered with its CRC-16 cut short or
    gone.  One byte further -- into the last subframe -- the frame still comes, through the ignored-failure path."""
    data, pcm, want = encoded(CONFIGS[0])
    full = oraclelib.flac_decode_file(data)
    for cut in (1, 2):
        o = oraclelib.flac_decode_file(data[:-cut])
        assert o["flags"] == 0 and o["n_frames"] == full["n_frames"]
        np.testing.assert_array_equal(o["pcm"], full["pcm"])
        np.testing.assert_array_equal(product_pcm(data[:-cut]), full["pcm"])
    o = oraclelib.flac_decode_file(data[:-3])
    assert o["flags"] & oraclelib.FLAC_F_IGNORED_FAILURE and o["n_frames"] == full["n_frames"]
    np.testing.assert_array_equal(product_pcm(data[:-3]), full["pcm"][:o["first_flag_sample"]])


def test_truncated_subframe_drops_the_frame_unless_it_is_the_last_subframe():
    """drflac.d:1591-1594 ignores what the sample decoders return; the NEXT subframe's header read then fails at the end of
    the data (:1575) and the frame is dropped -- unless the cut subframe was the frame's last: then the frame is delivered
    with what the decode buffer held.  Every cut of a stereo file shows one of the two."""
    data, pcm, want = encoded(CONFIGS[0])
    full = oraclelib.flac_decode_file(data)
    ends = np.concatenate([[0], np.cumsum(want[0]["block_size"].astype(np.int64) * 2)])     # samples delivered after k whole frames
    dropped = delivered = 0
    for cut in range(60, len(data) - 3, 131):
        o = oraclelib.flac_decode_file(data[:cut])
        whole = o["n_frames"] if not o["flags"] else o["n_frames"] - 1            # frames that were complete in the prefix
        clean = o["pcm"] if not o["flags"] else o["pcm"][:o["first_flag_sample"]]
        assert len(clean) == ends[whole]
        np.testing.assert_array_equal(clean, full["pcm"][:len(clean)])
        if o["flags"]:
            assert o["flags"] & oraclelib.FLAC_F_IGNORED_FAILURE and len(o["pcm"]) == ends[whole + 1]
            delivered += 1
        else:
            dropped += 1
        np.testing.assert_array_equal(product_pcm(data[:cut]), clean)               # the product stops at the cut frame either way
    assert dropped > 20 and delivered > 20, (dropped, delivered)


def damaged(base, rng):
    b = bytearray(base)
    kind = int(rng.integers(0, 5))
    if kind == 0:                                   # bit flips
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(42, len(b)))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:                                 # truncation (also inside subframes)
        b = b[:int(rng.integers(42, len(b)))]
    elif kind == 2:                                 # a run of one byte value
        pos = int(rng.integers(42, len(b) - 8))
        run = int(rng.integers(1, 8))
        b[pos:pos + run] = bytes([int(rng.integers(0, 256))]) * run
    elif kind == 3:                                 # deleted bytes
        pos = int(rng.integers(42, len(b)))
        del b[pos:pos + int(rng.integers(1, 16))]
    else:                                           # damage inside the STREAMINFO / metadata walk
        b[int(rng.integers(4, 42))] ^= 1 << int(rng.integers(0, 8))
    return bytes(b)


def test_product_parser_equals_the_oracle_parser_on_damaged_files():
    """1000 damaged files.  Identical samples, except where the reference itself leaves its defined ground:
      * AFGO_FLAC_F_IGNORED_FAILURE: a subframe's decode failed half way and the reference delivered the frame with what
        its decode buffer held (stale samples of earlier frames, or malloc'ed memory: AFGO_FLAC_F_UNINITIALISED).  The
        product ends the stream AT that frame: its output is the oracle's up to first_flag_sample, nothing else.
      * AFGO_FLAC_F_UNDEFINED: the oracle ended the stream where the reference would run an undefined operation; the
        product ends it there too (same samples)."""
    files = [encoded(c, s)[0] for s in range(2) for c in CONFIGS]
    rng = np.random.default_rng(20260510)
    counts = {"identical": 0, "stopped at the flagged frame": 0, "both refuse": 0}
    for it in range(1000):
        data = damaged(files[it % len(files)], rng)
        o = oraclelib.flac_decode_file(data)
        p = product_pcm(data)
        if isinstance(o, int):
            assert p is None or len(p) == 0, f"case {it}: the reference does not open the file, the product decodes {len(p)} samples"
            counts["both refuse"] += 1
            continue
        if p is None:
            p = np.zeros(0, np.int32)
        if len(p) == len(o["pcm"]):
            np.testing.assert_array_equal(p, o["pcm"], err_msg=f"case {it}")
            counts["identical"] += 1
        else:
            assert o["flags"] & oraclelib.FLAC_F_IGNORED_FAILURE, f"case {it}: {len(p)} samples against {len(o['pcm'])}, flags {o['flags']}"
            assert len(p) == o["first_flag_sample"], f"case {it}"
            np.testing.assert_array_equal(p, o["pcm"][:len(p)], err_msg=f"case {it}")
            counts["stopped at the flagged frame"] += 1
    assert counts["identical"] >= 500, counts
    print(counts)


# --------------------------------------------------------------------------------------------------------------- QOA
def qoa_file(n, channels, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    pcm = np.stack([(9000 * np.sin(0.02 * (c + 1) * t) + 800 * rng.standard_normal(n)) for c in range(channels)], 1)
    pcm = np.clip(np.round(pcm), -32768, 32767).astype(np.int16)
    return oraclelib.qoa_encode(pcm, 44100)


def product_qoa(data):
    try:
        frames = afgpu.qoa_parse(data)[0]                      # afg_qoa_parse: the product's host code
    except Exception:
        return None
    total = int(sum(int(f["samples"]) * int(f["channels"]) for f in frames))
    out = oraclelib.qoa_transform(frames, np.frombuffer(data, np.uint8), total, want_float=False)
    return out[0] if isinstance(out, tuple) else out


@pytest.mark.parametrize("n,channels", [(5120 * 2 + 333, 2), (20, 1), (5120, 3)])
def test_qoa_stream_layer_returns_the_encoder_reconstruction(n, channels):
    data, recon = qoa_file(n, channels, n)
    o = oraclelib.qoa_decode_file(bytes(data))
    assert (o["channels"], o["samplerate"], o["samples"]) == (channels, 44100, n)
    np.testing.assert_array_equal(o["pcm"].reshape(-1, channels), recon)
    np.testing.assert_array_equal(product_qoa(bytes(data)), o["pcm"])


def test_qoa_product_parser_equals_the_oracle_reader_on_damaged_files():
    files = [bytes(qoa_file(n, c, 5 * n + c)[0]) for n, c in ((5120 * 2 + 333, 2), (777, 1), (5120 + 40, 2))]
    rng = np.random.default_rng(77)
    same = refused = 0
    for it in range(600):
        b = bytearray(files[it % len(files)])
        kind = int(rng.integers(0, 4))
        if kind == 0:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            b = b[:int(rng.integers(0, len(b)))]
        elif kind == 2:                            # header fields of a frame
            fr = 8 + int(rng.integers(0, 2)) * (8 + 16 * b[8] + 256 * 8 * b[8]) if len(b) > 9000 else 8
            if fr + 8 <= len(b):
                b[fr + int(rng.integers(0, 8))] ^= 1 << int(rng.integers(0, 8))
        else:
            pos = int(rng.integers(0, len(b)))
            del b[pos:pos + int(rng.integers(1, 16))]
        data = bytes(b)
        o = oraclelib.qoa_decode_file(data)
        p = product_qoa(data)
        if isinstance(o, int) or len(o["pcm"]) == 0:
            assert p is None or len(p) == 0, f"case {it}"
            refused += 1
            continue
        assert p is not None, f"case {it}: the product refuses a file the reference reads"
        np.testing.assert_array_equal(p, o["pcm"], err_msg=f"case {it}")
        same += 1
    assert same >= 300, (same, refused)


def _product_vs_oracle(data):
    """afg_flac_parse's records through the restore oracle against the oracle front-end's delivery; -> (ok, frames)"""
    import afgpu
    o = oraclelib.flac_decode_file(data)
    want = None if isinstance(o, int) else o["pcm"]
    if want is not None and o["flags"] and o["first_flag_sample"] is not None:
        want = want[:o["first_flag_sample"]]                   # the product ends the stream at a stale-buffer frame (HISTORY.md 4)
    try:
        info, frames, subframes, res = afgpu.flac_parse(data)
        got = oraclelib.flac_transform(frames, subframes, res, info["out_samples"])
    except afgpu.AfgError:
        got = np.zeros(0, np.int32)
    want = np.zeros(0, np.int32) if want is None else want
    return len(got) == len(want) and np.array_equal(got, want), len(got)


def test_the_end_of_a_stream_as_the_references_rice_loop_reads_it():
    """drflac.d:1166-1236: the fused Rice loop fetches the next 32-bit line whenever a symbol reaches the end of the line its
    stop bit sits in -- so a symbol that ends on the LAST bit of the data fails, and a partial last line (1-3 bytes) fetched
    that way counts as a whole one, its missing bytes zeros.  Files cut short get there: tests/golden/soak_r05_cut_{a,b}.flac
    are the two of 2 000 damaged files (tools/soak_damaged.py, seed 13) on which the product's reader -- a plain positional
    one until then -- delivered a frame more, or less, than the reference; and every cut point in the last 300 bytes of two
    encoder-made files.  soak_r05_cut_c.flac (seed 41) is drflac__read_uint32's side of it (:834-856): a read that runs from
    a whole line into the partial last line and takes more than it holds pushes the count of consumed bits past 32, the
    unsigned "bits remaining" becomes enormous and every later read succeeds with zeros -- a verbatim subframe 12 000 bits
    longer than the file is delivered, padded with zeros."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for name in ("soak_r05_cut_a.flac", "soak_r05_cut_b.flac", "soak_r05_cut_c.flac"):
        ok, n = _product_vs_oracle(open(os.path.join(here, name), "rb").read())
        assert ok and n > 0, name
    import flac_bitstream as fb
    from test_flac_frontend import make_pcm
    rng = np.random.default_rng(17)
    for k in range(2):
        base, _ = fb.encode_file(make_pcm(int(rng.integers(5000, 9000)), 1 + k, 16, int(rng.integers(0, 1 << 30))), 16, 4096, orders=(8, 12, 2))
        for cut in range(len(base) - 300, len(base) + 1):
            ok, _ = _product_vs_oracle(base[:cut])
            assert ok, (k, cut, len(base))


def test_wasted_bits_are_counted_in_a_byte():
    """drflac.d:1563: wastedBitsPerSample = cast(ubyte)(cast(ubyte)count + 1) -- 276 zeros in front of the stop bit are 21
    wasted bits, not 277 (found by the soak's wider generators: a damaged 24-bit file; the product had refused the frame).
    A hand-made stream: one frame of 16 samples, one constant subframe with that count."""
    bits = []

    def put(v, n):
        bits.extend((v >> (n - 1 - i)) & 1 for i in range(n))
    # STREAMINFO: block sizes 16 / 16, frame sizes 0 / 0, 44100 Hz, 1 channel, 24 bits, 16 samples, MD5 0
    put(16, 16); put(16, 16); put(0, 24); put(0, 24); put(44100, 20); put(0, 3); put(23, 5); put(16, 36); put(0, 128)
    info = bytes(int("".join(map(str, bits[i:i + 8])), 2) for i in range(0, len(bits), 8))
    bits.clear()
    # frame header: sync, reserved, fixed blocking, block size code 6 (8-bit value follows), rate from STREAMINFO,
    # channel assignment 0 (mono), sample size from STREAMINFO, reserved, frame number 0, block size - 1, CRC-8
    put(0x3FFE, 14); put(0, 1); put(0, 1); put(6, 4); put(0, 4); put(0, 4); put(0, 3); put(0, 1); put(0, 8); put(15, 8); put(0xAB, 8)
    put(0x01, 8)                      # subframe: constant, wasted-bits flag
    put(0, 276); put(1, 1)            # the count
    put(0b101, 3)                     # 24 - 21 = 3 bits of constant: -3
    while len(bits) % 8: bits.append(0)
    put(0x1234, 16)                   # CRC-16 (not verified)
    frame = bytes(int("".join(map(str, bits[i:i + 8])), 2) for i in range(0, len(bits), 8))
    data = b"fLaC" + bytes([0x80, 0, 0, 34]) + info + frame
    o = oraclelib.flac_decode_file(data)
    assert not isinstance(o, int) and o["n_frames"] == 1 and o["flags"] == 0
    assert (o["pcm"] == (-3 << 21 << 8)).all() and len(o["pcm"]) == 16
    ok, n = _product_vs_oracle(data)
    assert ok and n == 16


def _stream_with_headers(channels, bps, min_bs, max_bs, total, frame_blobs, rate=44100):
    info = fb.streaminfo(rate, channels, bps, total, min_bs, max_bs)
    return b"fLaC" + fb.metadata_block(0, info, True) + b"".join(frame_blobs)


def test_frame_with_fewer_channels_than_streaminfo():
    """STREAMINFO says stereo, one frame in the middle is mono (its own header says so).  The reference decodes every frame by
    its own header and drflac_read_s32 walks frames by their own channel counts: whatever it delivers -- count and values --
    the product's parser + restore delivers (the product derived a length from STREAMINFO's channel count)."""
    st = make_pcm(256 * 3, 2, 16, 21)
    mono = make_pcm(256, 1, 16, 22)
    f2, s2, r2, _ = enc.encode(st, 16, 256, orders=(2, 8), use_fixed_every=1000)
    f1, s1, r1, _ = enc.encode(mono, 16, 256, orders=(2,), use_fixed_every=1000)
    blobs = [fb.write_frame(f2[0], s2, r2, 0, 44100, 16), fb.write_frame(f2[1], s2, r2, 1, 44100, 16),
             fb.write_frame(f1[0], s1, r1, 2, 44100, 16), fb.write_frame(f2[2], s2, r2, 3, 44100, 16)]
    for total in (256 * 4, 0):                                   # declared length, and none
        data = _stream_with_headers(2, 16, 256, 256, total, blobs)
        o = oraclelib.flac_decode_file(data)
        assert isinstance(o, dict) and o["n_frames"] >= 2
        want = o["pcm"] if not o["flags"] else o["pcm"][:o["first_flag_sample"]]
        got = product_pcm(data)
        assert got is not None and got.size == want.size, (got.size if got is not None else None, want.size)
        np.testing.assert_array_equal(got, want)


def test_streaminfo_with_zero_max_block():
    """max_block = 0 in STREAMINFO (a header no encoder writes): the oracle says what the reference's open and read make of it;
    the product agrees on open / refuse and on every delivered sample."""
    pcm = make_pcm(192 * 3, 2, 16, 23)
    f, s, r, _ = enc.encode(pcm, 16, 192, orders=(2, 8), use_fixed_every=1000)
    blobs = [fb.write_frame(f[i], s, r, i, 44100, 16) for i in range(len(f))]
    data = _stream_with_headers(2, 16, 0, 0, 192 * 3, blobs)
    o = oraclelib.flac_decode_file(data)
    got = product_pcm(data)
    if not isinstance(o, dict):
        assert got is None or got.size == 0
    else:
        want = o["pcm"] if not o["flags"] else o["pcm"][:o["first_flag_sample"]]
        assert got is not None and got.size == want.size
        np.testing.assert_array_equal(got, want)
