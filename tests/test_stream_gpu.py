"""End to end through the outer surface of the C ABI on the device: file bytes -> afg_open_from_memory /
afg_batch_decode -> interleaved floats, against (host front-end records -> oracle restore) and against the
PCM that was encoded.  Reference behaviour: stream.d:150-170, :295-412, :492-513 (FLAC read), :576-588 (QOA
read), qoa.d:810-851."""
import numpy as np
import pytest

import afgpu
import flac_bitstream as fb
import flac_ref_encoder as enc
import oraclelib
from test_flac_frontend import make_pcm

pytestmark = pytest.mark.gpu


def flac_expected(data):
    info, frames, subframes, res = afgpu.flac_parse(data)
    _, f32 = oraclelib.flac_transform(frames, subframes, res, info["out_samples"], want_float=True)
    return info, f32.reshape(-1, info["channels"])


def qoa_file(n, channels, rate, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    pcm = np.stack([9000 * np.sin(0.02 * (c + 1) * t) + 300 * rng.standard_normal(n) for c in range(channels)], 1)
    data, _ = oraclelib.qoa_encode(pcm.round().astype(np.int16), rate)
    frames, ch, _, total = afgpu.qoa_frames(data.tobytes())
    want = oraclelib.qoa_transform(frames, data, total * ch)[1].reshape(-1, ch)
    return data.tobytes(), want


def read_all(stream, channels, chunk):
    parts = []
    while True:
        buf = np.full(chunk * channels, np.nan, np.float32)
        got = stream.readSamplesFloat(buf)
        parts.append(buf[:got * channels])
        if got < chunk:
            break
    return np.concatenate(parts).reshape(-1, channels)


@pytest.mark.parametrize("channels,bps,block,n", [(2, 16, 4096, 4096 * 5 + 123), (2, 24, 1152, 1152 * 3), (1, 8, 576, 2000)])
def test_flac_stream(gpu, channels, bps, block, n):
    pcm = make_pcm(n, channels, bps, 21)
    data, _ = fb.encode_file(pcm, bps, block, sample_rate=48000, assignments=(enc.INDEPENDENT, enc.LEFT_SIDE, enc.MID_SIDE))
    info, want = flac_expected(data)
    s = afgpu.AudioStream()
    s.openFromMemory(data)
    assert not s.isError(), s.errorMessage()
    assert s.errorMessage() is None
    assert s.getFormat() == afgpu.FORMAT_FLAC and afgpu.FORMAT_NAMES[s.getFormat()] == "flac"
    assert s.getNumChannels() == channels and s.getSamplerate() == 48000.0 and s.getLengthInFrames() == n
    got = read_all(s, channels, 1000)
    assert got.shape == (n, channels)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))                 # bit-exact floats
    assert s.readSamplesFloat(np.zeros(channels * 4, np.float32)) == 0               # stream.d:498
    s.cleanUp()


def test_flac_declared_length_zero_reads_nothing(gpu):
    """STREAMINFO total = 0: _lengthInFrames = 0 and stream.d:498 returns 0 at position 0."""
    pcm = make_pcm(512, 2, 16, 4)
    frames, subframes, res, _ = enc.encode(pcm, 16, 256)
    data = fb.write_file(frames, subframes, res, 44100, 16, total_samples=0)
    s = afgpu.AudioStream()
    s.openFromMemory(data)
    assert not s.isError() and s.getLengthInFrames() == 0
    assert s.readSamplesFloat(np.zeros(64, np.float32)) == 0


def test_qoa_stream(gpu):
    data, want = qoa_file(5120 * 3 + 77, 2, 32000, 5)
    s = afgpu.AudioStream()
    s.openFromMemory(data)
    assert not s.isError(), s.errorMessage()
    assert s.getFormat() == afgpu.FORMAT_QOA and s.getNumChannels() == 2
    assert s.getSamplerate() == 32000.0 and s.getLengthInFrames() == 5120 * 3 + 77
    got = read_all(s, 2, 4099)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_batch_decode_mixed_formats_and_bad_files(gpu):
    files, wants = [], []
    for i in range(6):
        pcm = make_pcm(3000 + 517 * i, 1 + i % 2, 16, 30 + i)
        d, _ = fb.encode_file(pcm, 16, 1024, sample_rate=44100)
        files.append(d)
        wants.append(flac_expected(d)[1])
    for i in range(3):
        d, w = qoa_file(6000 + 1000 * i, 1 + i % 2, 44100, 40 + i)
        files.append(d)
        wants.append(w)
    files.insert(4, b"\x00" * 100)                 # junk in the middle must not poison the batch
    wants.insert(4, None)
    files.append(b"")
    wants.append(None)
    out = afgpu.batch_decode(files, n_threads=3)
    assert len(out) == len(files)
    for item, want, blob in zip(out, wants, files):
        if want is None:
            assert item["status"] != 0 and item["pcm"] is None
            assert item["message"] == "Cannot decode stream: unrecognized encoding."
            continue
        assert item["status"] == 0 and item["message"] is None
        assert item["format"] == (afgpu.FORMAT_FLAC if blob[:4] == b"fLaC" else afgpu.FORMAT_QOA)
        assert item["frames"] == len(want) and item["channels"] == want.shape[1]
        assert np.array_equal(item["pcm"].view(np.uint32), want.view(np.uint32))
    assert afgpu.batch_decode([]) == []


def test_flac_batch_mixes_declared_undeclared_and_misdeclared_lengths(gpu):
    """The batch path parses FLAC files of declared length straight into a staging plane sized from STREAMINFO; a file
    that declares nothing, fewer frames than it holds (more audio than the plane reserves) or more (truncated) must
    come out exactly as it does on its own."""
    files = []
    for i in range(5):
        pcm = make_pcm(2048 + 300 * i, 2, 16, 60 + i)
        frames, subframes, res, _ = enc.encode(pcm, 16, 512)
        total = None
        if i == 1: total = 0                                   # undeclared: own buffer in pass 1
        if i == 2: total = 700                                 # under-declared: the staged parse overflows its plane
        if i == 3: total = 100000                              # over-declared (as if truncated)
        files.append(fb.write_file(frames, subframes, res, 44100, 16, total_samples=total))
    alone = [afgpu.batch_decode([f], n_threads=1)[0] for f in files]
    for order in (files, files[::-1], [files[0], files[4]], [files[0], files[2], files[4]]):
        got = afgpu.batch_decode(order, n_threads=4)
        for g, f in zip(got, order):
            a = alone[files.index(f)]
            assert g["status"] == a["status"] == 0 and g["frames"] == a["frames"] and g["frames"] > 0
            assert np.array_equal(g["pcm"].view(np.uint32), a["pcm"].view(np.uint32))
    want = flac_expected(files[0])[1]
    assert np.array_equal(alone[0]["pcm"].view(np.uint32), want.view(np.uint32))


# ---- MP3: file bytes -> host front-end -> device transform -> what mp3dec_ex_read delivers --------------------
import os  # noqa: E402

MP3_FIXTURE = os.path.join(os.path.dirname(__file__), "golden", "mathjax_invalid_keypress.mp3")


def test_mp3_stream(gpu):
    data = open(MP3_FIXTURE, "rb").read()
    want = oraclelib.mp3_decode_file(data)
    s = afgpu.AudioStream()
    s.openFromMemory(data)
    assert not s.isError(), s.errorMessage()
    assert s.getFormat() == afgpu.FORMAT_MP3 and s.getNumChannels() == 2 and s.getSamplerate() == 44100.0
    assert s.getLengthInFrames() == want["declared_samples"] // 2 == 23087            # stream.d:1737
    got = read_all(s, 2, 777)
    assert got.shape == (23087, 2)
    assert np.array_equal(got.view(np.uint32), want["pcm"].reshape(-1, 2).view(np.uint32))      # bit-exact
    assert s.readSamplesFloat(np.zeros(8, np.float32)) == 0


def test_mp3_variants_in_a_batch(gpu):
    d = open(MP3_FIXTURE, "rb").read()
    body = d[45:]
    files = [d, body[209:], bytes(300) + body, d[:5000], d + b"TAG" + bytes(125)]
    rng = np.random.default_rng(11)
    for _ in range(6):                                   # damaged copies: resynchronisation splits them into several runs
        v = bytearray(d)
        pos = int(rng.integers(600, len(v) - 2500))
        del v[pos:pos + int(rng.integers(1, 200))]
        files.append(bytes(v))
    pcm = make_pcm(3000, 2, 16, 5)
    flac, _ = fb.encode_file(pcm, 16, 1024)
    files.insert(3, flac)                                # formats can be mixed freely
    out = afgpu.batch_decode(files, n_threads=4)
    for blob, item in zip(files, out):
        if blob[:4] == b"fLaC":
            assert item["format"] == afgpu.FORMAT_FLAC and item["frames"] == 3000
            continue
        want = oraclelib.mp3_decode_file(blob)
        assert item["status"] == 0 and item["format"] == afgpu.FORMAT_MP3
        assert item["frames"] * item["channels"] == len(want["pcm"])
        if item["frames"]:
            assert np.array_equal(item["pcm"].reshape(-1).view(np.uint32), want["pcm"].view(np.uint32))


# ---- Ogg Vorbis ---------------------------------------------------------------------------------------------
OGG_FIXTURE = os.path.join(os.path.dirname(__file__), "golden", "mathjax_invalid_keypress.ogg")


def test_ogg_stream(gpu):
    data = open(OGG_FIXTURE, "rb").read()
    want = oraclelib.vorbis_file_pcm(oraclelib.vorbis_decode_file(data))
    s = afgpu.AudioStream()
    s.openFromMemory(data)
    assert not s.isError(), s.errorMessage()
    assert s.getFormat() == afgpu.FORMAT_OGG and s.getNumChannels() == 2 and s.getSamplerate() == 44100.0
    assert s.getLengthInFrames() == 22050                                            # stream.d:1696
    got = read_all(s, 2, 1000)
    assert got.shape == want.shape == (22050, 2)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))                 # bit-exact


def test_all_formats_in_one_batch(gpu):
    ogg = open(OGG_FIXTURE, "rb").read()
    mp3 = open(MP3_FIXTURE, "rb").read()
    flac, _ = fb.encode_file(make_pcm(5000, 2, 16, 8), 16, 1024)
    qoa, qoa_want = qoa_file(7000, 2, 44100, 9)
    cut = ogg[:len(ogg) - 300]                                                       # a cut Ogg file still yields its head
    files = [ogg, mp3, flac, qoa, cut, ogg, b"junk" * 100]
    out = afgpu.batch_decode(files, n_threads=4)
    assert [o["format"] for o in out[:6]] == [afgpu.FORMAT_OGG, afgpu.FORMAT_MP3, afgpu.FORMAT_FLAC, afgpu.FORMAT_QOA,
                                              afgpu.FORMAT_OGG, afgpu.FORMAT_OGG]
    assert out[6]["status"] != 0
    want_ogg = oraclelib.vorbis_file_pcm(oraclelib.vorbis_decode_file(ogg))
    for k in (0, 5):
        assert np.array_equal(out[k]["pcm"].view(np.uint32), want_ogg.view(np.uint32))
    from test_vorbis_frontend import cut_at_window_break
    want_cut = oraclelib.vorbis_file_pcm(cut_at_window_break(oraclelib.vorbis_decode_file(cut, seek_clears_eof=True)))
    assert out[4]["frames"] == len(want_cut) > 5000
    assert np.array_equal(out[4]["pcm"].view(np.uint32), want_cut.view(np.uint32))
    assert np.array_equal(out[1]["pcm"].reshape(-1).view(np.uint32), oraclelib.mp3_decode_file(mp3)["pcm"].view(np.uint32))
    assert np.array_equal(out[3]["pcm"].view(np.uint32), qoa_want.view(np.uint32))
    assert out[2]["frames"] == 5000


def test_synthetic_vorbis_streams_end_to_end(gpu):
    """Generated streams (tests/vorbis_bitstream.py) through the whole path: mono .. 6 channels, several block-size
    pairs (wave kernel and general kernel), all residue types."""
    import vorbis_bitstream as vb
    files, wants = [], []
    for k, (ch, bs) in enumerate([(1, (256, 2048)), (2, (256, 2048)), (2, (512, 512)), (3, (256, 1024)), (6, (1024, 4096)),
                                  (2, (2048, 8192))]):
        d = vb.make_file(900 + k, channels=ch, bs=bs, n_packets=16, residue_types=[(0, 1), (1, 2), (2, 0)][k % 3])
        rec = oraclelib.vorbis_decode_file(d)
        assert rec is not None and len(rec["pflags"]) >= 10
        files.append(d)
        wants.append(oraclelib.vorbis_file_pcm(rec))
    out = afgpu.batch_decode(files, n_threads=2)
    for item, want in zip(out, wants):
        assert item["status"] == 0 and item["format"] == afgpu.FORMAT_OGG
        assert item["frames"] == len(want) and item["channels"] == want.shape[1]
        if len(want):
            rms = float(np.sqrt(np.mean((item["pcm"].astype(np.float64) - want) ** 2)))
            assert rms <= 1e-5                                                   # north-star tolerance, absolute (calibrated generators)
            assert np.array_equal(item["pcm"].view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("which", ["flac", "qoa", "mp3", "ogg"])
def test_seek_and_tell_invariants(gpu, which):
    """the checks of the reference's examples/transcode additionalTests (main.d:93-160)"""
    if which == "flac":
        data, _ = fb.encode_file(make_pcm(5000, 2, 16, 3), 16, 1024)
    elif which == "qoa":
        data, _ = qoa_file(6000, 2, 44100, 3)
    else:
        data = open(MP3_FIXTURE if which == "mp3" else OGG_FIXTURE, "rb").read()
    s = afgpu.AudioStream()
    s.openFromMemory(data)
    assert not s.isError() and s.canSeek()
    ch, n = s.getNumChannels(), s.getLengthInFrames()
    whole = read_all(s, ch, 4096)
    assert len(whole) == n
    assert s.seekPosition(0) and s.tellPosition() == 0
    assert not s.seekPosition(n + 1) and s.tellPosition() == 0           # past the end: refused, a no-op
    assert not s.seekPosition(-1) and s.tellPosition() == 0
    assert s.seekPosition(n // 2) and s.tellPosition() == n // 2
    buf = np.zeros(64 * ch, np.float32)
    assert s.readSamplesFloat(buf) == 64 and s.tellPosition() == n // 2 + 64
    assert np.array_equal(buf.reshape(-1, ch), whole[n // 2:n // 2 + 64])   # sample-accurate
    assert s.seekPosition(n - 1) and s.tellPosition() == n - 1
    assert s.readSamplesFloat(np.zeros(16 * ch, np.float32)) == 1        # exactly one frame is left
    assert s.seekPosition(n) and s.readSamplesFloat(np.zeros(16 * ch, np.float32)) == 0


def test_batch_decode_from_several_host_threads(gpu):
    """Distinct calls share no mutable state but the pools (page-locked buffers, helper threads, table caches): four host
    threads decoding different mixed batches at once get what they get alone."""
    import threading
    rng = np.random.default_rng(77)
    mp3 = open(MP3_FIXTURE, "rb").read()
    ogg = open(OGG_FIXTURE, "rb").read()
    pool = [mp3, ogg, mp3[:len(mp3) * 2 // 3], ogg[:len(ogg) * 3 // 4]]
    for i in range(3):
        pool.append(fb.encode_file(make_pcm(1500 + 400 * i, 1 + i % 2, 16, 90 + i), 16, 512)[0])
        pool.append(qoa_file(5120 + 333 * i, 1 + i % 2, 44100, 95 + i)[0])
    batches = [[pool[int(k)] for k in rng.integers(0, len(pool), 24)] for _ in range(4)]
    alone = [afgpu.batch_decode(b, n_threads=4) for b in batches]
    got = [None] * len(batches)
    errs = []

    def work(k):
        try:
            for _ in range(3):
                got[k] = afgpu.batch_decode(batches[k], n_threads=4)
        except Exception as e:                                   # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(k,)) for k in range(len(batches))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for a, g in zip(alone, got):
        assert len(a) == len(g)
        for x, y in zip(a, g):
            assert (x["status"], x["frames"], x["channels"], x["format"]) == (y["status"], y["frames"], y["channels"], y["format"])
            if x["pcm"] is not None:
                assert np.array_equal(x["pcm"].view(np.uint32), y["pcm"].view(np.uint32))


def test_flac_lossless_round_trip_over_the_parameter_space(gpu):
    """encode (test encoder) -> batch decode (HIP) returns the PCM exactly, for 1..8 channels, 8..24 bits and block
    sizes from 16 to 4608 -- checked against the input itself (stream.d:505-511 conversion), not against the oracle."""
    rng = np.random.default_rng(2024)
    cases, files = [], []
    for k in range(14):
        ch = int(rng.integers(1, 9))
        bps = int(rng.choice([8, 12, 16, 20, 24]))
        block = int(rng.choice([16, 100, 576, 1024, 4096, 4608]))
        n = int(block * rng.integers(1, 4) + rng.integers(0, block))
        pcm = make_pcm(n, ch, bps, 300 + k)
        kw = dict(orders=(0, 1, 2, 4, 8, 12, 32)) if block >= 64 else dict(orders=(0, 1, 2, 4))
        if ch == 2:
            kw["assignments"] = (enc.INDEPENDENT, enc.LEFT_SIDE, enc.RIGHT_SIDE, enc.MID_SIDE)
        files.append(fb.encode_file(pcm, bps, block, sample_rate=44100, **kw)[0])
        cases.append((pcm, bps))
    got = afgpu.batch_decode(files, n_threads=4)
    for g, (pcm, bps) in zip(got, cases):
        assert g["status"] == 0 and g["frames"] == len(pcm) and g["channels"] == pcm.shape[1]
        s32 = (pcm.astype(np.int64) << (32 - bps)).astype(np.int32)
        want = (s32.astype(np.float64) * (1.0 / 2147483647.0)).astype(np.float32)
        assert np.array_equal(g["pcm"].reshape(-1, pcm.shape[1]).view(np.uint32), want.view(np.uint32))


# ---- chunked reading: the stream decodes as the caller pulls (stream.d:429-637) -------------------------------------
def long_files():
    """One file per format, each several decode chunks long (64 MP3 frames / 64 Vorbis packets / 16 FLAC or QOA frames)."""
    import mp3_bitstream as mb
    import vorbis_bitstream as vb
    rng = np.random.default_rng(77)
    n = 4096 * 50 + 321
    t = np.arange(n)
    pcm = np.stack([9000 * np.sin(0.01 * t) + 800 * rng.standard_normal(n), 7000 * np.sin(0.013 * t + 1) + 800 * rng.standard_normal(n)], 1)
    out = {"flac": fb.encode_file(pcm.round().astype(np.int64), 16, 4096, orders=(8, 12))[0],
           "qoa": oraclelib.qoa_encode(pcm[:5120 * 40 + 99].round().astype(np.int16), 44100)[0].tobytes(),
           "mp3": mb.make_file(123, n_frames=300, version="mpeg1", sr=0, mode="ms", bitrate_index=9)[0],
           "ogg": vb.make_file(124, n_packets=300)}
    d = bytearray(out["mp3"])
    del d[40000:40123]                                   # a damaged copy: the decoder resynchronises (a new run) inside a chunk
    out["mp3_damaged"] = bytes(d)
    return out


@pytest.mark.parametrize("kind", ["flac", "qoa", "mp3", "mp3_damaged", "ogg"])
def test_chunked_reads_equal_the_batch_decode(gpu, kind):
    data = long_files()[kind]
    want = afgpu.batch_decode([data])[0]
    assert want["status"] == 0 and want["frames"] > 20000
    ch = want["channels"]
    for chunk in (1024, 777, 100000):                    # the example's read size, an odd one, one larger than a decode chunk
        s = afgpu.AudioStream()
        s.openFromMemory(data)
        assert not s.isError(), s.errorMessage()
        assert s.getNumChannels() == ch and s.getSamplerate() == want["samplerate"]
        got = read_all(s, ch, chunk)
        assert not s.isError()
        assert got.shape == (want["frames"], ch)
        assert np.array_equal(got.view(np.uint32), want["pcm"].view(np.uint32))
        assert s.tellPosition() == want["frames"]
        s.cleanUp()


@pytest.mark.parametrize("kind", ["flac", "qoa", "mp3", "ogg"])
def test_seeks_across_decode_chunks(gpu, kind):
    """seekPosition / tellPosition / readSamplesFloat stay consistent when seeks jump forwards and backwards over several
    decode chunks (examples/transcode additionalTests, main.d:93-160, on a stream that is not held decoded)."""
    data = long_files()[kind]
    want = afgpu.batch_decode([data])[0]
    ch, total = want["channels"], want["frames"]
    s = afgpu.AudioStream()
    s.openFromMemory(data)
    length = s.getLengthInFrames()
    rng = np.random.default_rng(5)
    for target in [total // 2, 10, total - 7, total // 3, 0, total // 3 + 1, min(length, total)]:
        assert s.seekPosition(int(target))
        assert s.tellPosition() == min(target, total)
        buf = np.zeros(500 * ch, np.float32)
        got = s.readSamplesFloat(buf)
        assert got == min(500, total - min(target, total))
        assert np.array_equal(buf[:got * ch].view(np.uint32), want["pcm"][target:target + got].reshape(-1).view(np.uint32))
        assert s.tellPosition() == min(target, total) + got
    assert not s.seekPosition(-1) and not s.seekPosition(int(length) + 1)
    s.cleanUp()


@pytest.mark.parametrize("seed", [5, 13, 30, 45, 52])
def test_vorbis_streams_whose_delivery_starts_late(gpu, seed):
    """Streams whose first delivered frame is not the start of a packet's output (a long first block with a short next
    window: the deferred discard of stb_vorbis2.d:2551-2560) or comes after packets that deliver nothing: the stream API,
    the batch entry and the oracle agree sample for sample (the result plane offset of such a file is not zero)."""
    import vorbis_bitstream as vb
    data = vb.make_file(seed)
    rec = afgpu.vorbis_parse(data)
    first = next(i for i, c in enumerate(rec["take_count"]) if c > 0)
    assert rec["take_from"][first] > 0 or first > 1
    want = oraclelib.vorbis_file_pcm(oraclelib.vorbis_decode_file(data))
    item = afgpu.batch_decode([data])[0]
    assert item["status"] == 0 and item["frames"] == len(want)
    assert np.array_equal(item["pcm"].view(np.uint32), want.view(np.uint32))
    for chunk in (100, 4096):
        s = afgpu.AudioStream()
        s.openFromMemory(data)
        assert not s.isError(), s.errorMessage()
        got = read_all(s, s.getNumChannels(), chunk)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        if s.getLengthInFrames() > 0:                      # (a stream without a last-page granule has no length: no seeking, as in the reference)
            assert s.seekPosition(7) and s.tellPosition() == 7
            buf = np.zeros(50 * want.shape[1], np.float32)
            assert s.readSamplesFloat(buf) == min(50, len(want) - 7)
            assert np.array_equal(buf[:min(50, len(want) - 7) * want.shape[1]].view(np.uint32), want[7:57].reshape(-1).view(np.uint32))
        else:
            assert not s.seekPosition(7)
        s.cleanUp()
