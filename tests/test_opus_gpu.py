"""Ogg Opus (CELT-only) end to end on the device through the outer surface of the C ABI: file bytes -> afg_batch_decode /
afg_open_from_memory -> interleaved floats, against the oracle chain (oracle/opus_frontend.c records ->
afgo_celt_transform -> gain -> afgo_opus_output, clipped to the declared length).

Reference behaviour: stream.d:429-487 (Opus read loop, length clamp, int16 / 32767), :1596-1614 (Opus is probed first,
48 kHz, length = smpduration), dopus.d:6688-6691 (gain), :8062-8110 (readFrame), :3680-3702 (transform)."""
import os

import numpy as np
import pytest

import afgpu
import opus_bitstream as ob
import oraclelib
from test_stream_gpu import read_all

pytestmark = pytest.mark.gpu


def expected(data):
    rec = oraclelib.opus_decode_file(data)
    assert not isinstance(rec, int)
    return rec, oraclelib.opus_file_pcm(rec)


def same_pcm(got, want):
    assert got.shape == want.shape
    # the device transform follows the oracle's expression trees: after the int16 round trip the match is exact
    bad = np.nonzero(got.view(np.uint32).reshape(-1) != want.view(np.uint32).reshape(-1))[0]
    assert len(bad) == 0, (f"{len(bad)} of {got.size} samples differ, first at {int(bad[0])}, last at {int(bad[-1])}; "
                           f"got {got.reshape(-1)[bad[:12]]!r} want {want.reshape(-1)[bad[:12]]!r} at {bad[:12]!r}")


def test_batch_of_random_celt_files(gpu):
    rng = np.random.default_rng(21)
    files, wants = [], []
    for k in range(24):
        ch = 1 + k % 2
        # (a negative *header* gain reads as a large positive one in the reference; the R128 tag is how a file gets quieter.
        # The random payloads decode to ~70 dB over full scale: -78 dB brings them inside the int16 range)
        data, _ = ob.random_celt_file(rng, ch, int(rng.integers(1, 70)), preskip=int(rng.integers(0, 121)), gain=(0, 0, 700)[k % 3],
                                      comments=(b"R128_TRACK_GAIN=-20000",) if k % 3 == 1 else ())
        files.append(data)
        wants.append(expected(data))
    res = afgpu.batch_decode(files)
    for k, (r, (rec, pcm)) in enumerate(zip(res, wants)):
        print("file", k, "channels", rec["channels"], "frames", len(rec["frames"]), "gain_i", rec["gain_i"],
              "sizes", sorted(set(int(x) for x in rec["frames"]["frame_size"])))
        assert r["status"] == 0, r["message"]
        assert r["format"] == afgpu.FORMAT_OPUS and r["samplerate"] == 48000.0 and r["channels"] == rec["channels"]
        assert r["frames"] == len(pcm) == min(rec["pcm_frames"], rec["declared_frames"])
        same_pcm(r["pcm"], pcm)
    # quiet files exercise the conversion away from the saturation rails too
    quiet = [r for r, (rec, _) in zip(res, wants) if rec["gain_i"] < 0]
    assert any(np.abs(r["pcm"]).max() < 1.0 for r in quiet)


def test_opus_next_to_the_other_formats_and_bad_files(gpu):
    import mp3_bitstream as mb
    import vorbis_bitstream as vb
    rng = np.random.default_rng(22)
    opus, _ = ob.random_celt_file(rng, 2, 30, preskip=100)
    silk = ob.ogg_opus([ob.toc(3, True, 0) + rng.bytes(50)] * 3, 2, preskip=0)
    broken = ob.ogg_opus([ob.packet(rng, 31, True, 0, sizes=[90]), ob.toc(31, True, 1) + bytes(7)], 2, preskip=0)
    files = [opus, vb.make_file(5, n_packets=20), silk, mb.make_file(6, n_frames=12)[0], broken, b"OggS" + bytes(100)]
    res = afgpu.batch_decode(files)
    assert [r["format"] for r in res[:2]] == [afgpu.FORMAT_OPUS, afgpu.FORMAT_OGG]
    same_pcm(res[0]["pcm"], expected(opus)[1])
    assert res[1]["status"] == 0 and res[3]["status"] == 0 and res[3]["format"] == afgpu.FORMAT_MP3
    assert res[2]["status"] != 0 and "SILK" in res[2]["message"] and res[2]["pcm"] is None
    assert res[4]["status"] != 0 and res[4]["pcm"] is None                      # a packet that cannot be framed: the file is an error
    assert res[5]["status"] != 0


@pytest.mark.parametrize("channels", [1, 2])
def test_stream_reads_in_chunks_equal_the_batch_decode(gpu, channels):
    rng = np.random.default_rng(23 + channels)
    data, _ = ob.random_celt_file(rng, channels, 400, preskip=312, comments=(b"R128_TRACK_GAIN=-19000",))   # several decode chunks of 64 packets
    rec, want = expected(data)
    assert len(want) > 60000
    for chunk in (1024, 777, 200000):
        s = afgpu.AudioStream()
        s.openFromMemory(data)
        assert not s.isError(), s.errorMessage()
        assert s.getFormat() == afgpu.FORMAT_OPUS and s.getNumChannels() == channels and s.getSamplerate() == 48000.0
        assert s.getLengthInFrames() == rec["declared_frames"]
        got = read_all(s, channels, chunk)
        assert not s.isError()
        same_pcm(got, want)
        assert s.tellPosition() == len(want)
        buf = np.zeros(16 * channels, np.float32)
        assert s.readSamplesFloat(buf) == 0                                       # at the declared end: nothing more (stream.d:439-442)
        s.cleanUp()


def test_declared_length_longer_than_the_audio(gpu):
    rng = np.random.default_rng(25)
    pkts = [ob.packet(rng, 31, True, 0, sizes=[70]) for _ in range(20)]
    data = ob.ogg_opus(pkts, 2, preskip=0, trim=-5000)                            # the last page claims 5000 frames too many
    rec, want = expected(data)
    assert rec["declared_frames"] == 20 * 960 + 5000 and len(want) == 20 * 960
    s = afgpu.AudioStream()
    s.openFromMemory(data)
    got = read_all(s, 2, 4096)
    same_pcm(got, want)
    assert not s.isError()
    s.cleanUp()


def test_seek_and_tell(gpu):
    rng = np.random.default_rng(26)
    data, _ = ob.random_celt_file(rng, 2, 300, preskip=100)
    _, want = expected(data)
    total = len(want)
    s = afgpu.AudioStream()
    s.openFromMemory(data)
    length = s.getLengthInFrames()
    for target in [total // 2, 10, total - 7, total // 3, 0, total // 3 + 1, total]:
        assert s.seekPosition(int(target))
        assert s.tellPosition() == target
        buf = np.zeros(500 * 2, np.float32)
        got = s.readSamplesFloat(buf)
        assert got == min(500, total - target)
        assert np.array_equal(buf[:got * 2].view(np.uint32), want[target:target + got].reshape(-1).view(np.uint32))
    assert not s.seekPosition(-1) and not s.seekPosition(int(length) + 1)
    s.cleanUp()


def test_stream_reports_the_failing_packet_like_the_reference(gpu):
    """readFrame fails -> the read that reaches the packet sets the error and returns 0 (stream.d:452-456); what was read
    before stays valid"""
    rng = np.random.default_rng(27)
    good = [ob.packet(rng, 31, True, 0, sizes=[90]) for _ in range(100)]
    data = ob.ogg_opus(good[:70] + [ob.toc(31, True, 1) + bytes(7)] + good[70:], 2, preskip=0)
    clean = ob.ogg_opus(good[:70], 2, preskip=0)
    _, want = expected(clean)
    s = afgpu.AudioStream()
    s.openFromMemory(data)
    assert not s.isError()
    parts = []
    while True:
        buf = np.zeros(4096 * 2, np.float32)
        got = s.readSamplesFloat(buf)
        if s.isError():
            assert got == 0
            break
        parts.append(buf[:got * 2].copy())
        assert got == 4096
    got = np.concatenate(parts).reshape(-1, 2)
    assert len(got) == (70 * 960 // 4096) * 4096
    same_pcm(got, want[:len(got)])
    assert "initialization" in s.errorMessage()
    s.cleanUp()


def test_silk_file_is_refused_at_open(gpu):
    rng = np.random.default_rng(28)
    s = afgpu.AudioStream()
    s.openFromMemory(ob.ogg_opus([ob.toc(9, False, 0) + rng.bytes(30)] * 4, 1, preskip=0))
    assert s.isError() and "SILK" in s.errorMessage()
    s.cleanUp()


def test_output_gain_kernel_matches_the_oracle(gpu):
    import torch
    rng = np.random.default_rng(29)
    for n in (1, 3, 4, 1023, 4099):
        x = np.concatenate([rng.uniform(-1.5, 1.5, n).astype(np.float32), np.float32([0.5 / 32768, -0.5 / 32768, 1.0, -1.0, -500.0, -800.0, 900.0])])
        for gain_i in (-2560, 333, 0):
            g = np.float32(np.exp2(3.32192809488736234787 * (gain_i / (20.0 * 256))))          # (0: a multiply by 1.0f)
            want_i, want_f = oraclelib.opus_output(x * g)
            d_in = torch.from_numpy(x).to(gpu)
            d_i = torch.zeros(len(x), dtype=torch.int16, device=gpu)
            d_f = torch.zeros(len(x), dtype=torch.float32, device=gpu)
            afgpu.opus_output(len(x), d_in, d_i, d_f, gain=float(g))
            torch.cuda.synchronize()
            assert np.array_equal(d_i.cpu().numpy(), want_i)
            assert np.array_equal(d_f.cpu().numpy().view(np.uint32), want_f.view(np.uint32))
            # unaligned views take the scalar path
            if len(x) > 8:
                d_f2 = torch.zeros(len(x), dtype=torch.float32, device=gpu)
                afgpu.opus_output(len(x) - 1, d_in[1:], None, d_f2[1:], gain=float(g))
                torch.cuda.synchronize()
                assert np.array_equal(d_f2.cpu().numpy()[1:].view(np.uint32), want_f[1:].view(np.uint32))


# one int16 step on the float API's scale: int16 / 32767.0f values near full scale are 6e-8 apart, so neighbours differ by
# 1 / 32767 give or take that
ONE_STEP = 1 / 32767 + 1.2e-7


@pytest.mark.numeric_tolerance
def test_default_numeric_mode_end_to_end(gpu):
    """The product's default numeric mode (AFG_NUMERIC_TOLERANCE, csrc/celt_walk.hip) through the whole path -- batch decode
    and chunked stream reads (64-packet chunks with carried state) of files at programme level: what comes out is the
    reference's int16 / 32767 except for rare samples that the float transform's 1e-7 puts on the other side of a rounding
    boundary -- one step of 1/32767, fewer than 1 % of the samples, 1e-5 RMS (north_star)."""
    rng = np.random.default_rng(61)
    files, wants = [], []
    for k in range(10):
        ch = 1 + k % 2
        # -78 dB through the R128 tag brings the random payloads (~70 dB over full scale) inside the int16 range
        data, _ = ob.random_celt_file(rng, ch, int(rng.integers(80, 260)), preskip=int(rng.integers(0, 313)),
                                      comments=(b"R128_TRACK_GAIN=-20000",))
        files.append(data)
        wants.append(expected(data))
    assert afgpu.get_numeric_mode() == afgpu.NUMERIC_TOLERANCE
    res = afgpu.batch_decode(files)
    total = flips = 0
    for r, (rec, pcm), data in zip(res, wants, files):
        assert r["status"] == 0, r["message"]
        assert r["frames"] == len(pcm)
        for got in (r["pcm"],):
            d = np.abs(got.astype(np.float64) - pcm)
            assert d.max() <= ONE_STEP, d.max()
            assert np.sqrt(np.mean(d ** 2)) <= 1e-5
            total += d.size
            flips += int((d > 0).sum())
        if np.abs(pcm).max() < 1.0:
            assert np.abs(r["pcm"]).max() < 1.0 + 1e-6
        # the same file pulled through the stream surface in odd-sized reads: chunked decoding with the carry state
        s = afgpu.AudioStream()
        s.openFromMemory(data)
        got = read_all(s, rec["channels"], 3001)
        assert not s.isError(), s.errorMessage()
        s.cleanUp()
        d = np.abs(got.astype(np.float64) - pcm)
        assert got.shape == pcm.shape and d.max() <= ONE_STEP and np.sqrt(np.mean(d ** 2)) <= 1e-5
        flips += int((d > 0).sum())
        total += d.size
    print("int16 flip rate", flips / total)
    assert flips / total < 0.01


def encoder_made_files():
    import os
    v = np.load(os.path.join(os.path.dirname(__file__), "golden", "independent_webaudio.npz"))
    return [v[f"opus_enc{k}_file"].tobytes() for k in range(3)], [v[f"opus_enc{k}_pcm"] for k in range(3)]


def test_encoder_made_files_exact_mode(gpu):
    """Three clips from a real Opus encoder (Chromium's MediaRecorder: WebRTC's libopus, CELT-only fullband, code-3 packets of
    three 20 ms frames; tests/golden/make_independent.py), transient and stationary, stereo and mono: the product's decode
    is the oracle's bit for bit, batch and chunked stream reads alike -- and within 2e-3 of libopus' own decode of them."""
    files, libopus = encoder_made_files()
    res = afgpu.batch_decode(files)
    for r, data, lo in zip(res, files, libopus):
        rec, pcm = expected(data)
        assert r["status"] == 0 and r["format"] == afgpu.FORMAT_OPUS and r["channels"] == rec["channels"], r["message"]
        same_pcm(r["pcm"], pcm)
        st = afgpu.AudioStream()
        st.openFromMemory(data)
        assert not st.isError(), st.errorMessage()
        same_pcm(read_all(st, rec["channels"], 777), pcm)
        st.cleanUp()
        d = r["pcm"].astype(np.float64) - lo
        assert np.sqrt(np.mean(d ** 2)) <= 2e-3 * np.sqrt(np.mean(lo.astype(np.float64) ** 2))


@pytest.mark.numeric_tolerance
def test_encoder_made_files_default_mode(gpu):
    files, _ = encoder_made_files()
    assert afgpu.get_numeric_mode() == afgpu.NUMERIC_TOLERANCE
    for r, data in zip(afgpu.batch_decode(files), files):
        rec, pcm = expected(data)
        assert r["status"] == 0 and r["frames"] == len(pcm)
        d = np.abs(r["pcm"].astype(np.float64) - pcm)
        assert d.max() <= ONE_STEP and np.sqrt(np.mean(d ** 2)) <= 1e-5 and (d > 0).mean() < 0.01


@pytest.mark.numeric_tolerance
def test_batch_composition_does_not_change_a_files_samples(gpu):
    """A stereo file decodes to the same bits alone, behind a mono file and behind two (the transform stage walks the two
    channels of a stream in one wavefront when they sit on an even / odd pair of channel sequences: afg_batch_decode keeps
    stereo files there with an empty sequence after an odd number of mono ones; found by tools/soak_damaged.py)."""
    rng = np.random.default_rng(77)
    stereo = [ob.random_celt_file(rng, 2, 14, pcm_rms=0.05)[0] for _ in range(2)]
    mono = [ob.random_celt_file(rng, 1, 9, pcm_rms=0.05)[0] for _ in range(3)]
    alone = [afgpu.batch_decode([d])[0]["pcm"] for d in stereo]
    mono_alone = [afgpu.batch_decode([d])[0]["pcm"] for d in mono]
    for batch, where, monos in (([mono[0], stereo[0]], {1: 0}, {0: 0}), ([mono[0], mono[1], stereo[1]], {2: 1}, {0: 0, 1: 1}),
                                ([stereo[0], mono[2], stereo[1], mono[0], mono[1], stereo[0]], {0: 0, 2: 1, 5: 0}, {1: 2, 3: 0, 4: 1})):
        out = afgpu.batch_decode(batch)
        for i, k in where.items():
            assert out[i]["status"] == 0
            assert np.array_equal(out[i]["pcm"].view(np.uint32), alone[k].view(np.uint32)), (len(batch), i)
        for i, k in monos.items():
            assert np.array_equal(out[i]["pcm"].view(np.uint32), mono_alone[k].view(np.uint32)), (len(batch), i)


@pytest.mark.numeric_tolerance
def test_the_soak_pair_that_found_it(gpu):
    """tests/golden/soak_r05_{mono,stereo}.opus: two damaged generated files of one soak batch (tools/soak_damaged.py, round 67
    of seed 2024).  Behind the mono one the stereo file's channels sat on an odd / even pair of sequences, were walked
    one at a time and came out one int16 step away from the file decoded alone, at one sample."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    mono = open(os.path.join(here, "soak_r05_mono.opus"), "rb").read()
    stereo = open(os.path.join(here, "soak_r05_stereo.opus"), "rb").read()
    alone = afgpu.batch_decode([stereo])[0]["pcm"]
    for batch, at in (([mono, stereo], 1), ([mono, mono, stereo], 2), ([stereo, mono, stereo], 2), ([mono, stereo, mono, stereo], 3)):
        got = afgpu.batch_decode(batch)[at]
        assert got["status"] == 0 and np.array_equal(got["pcm"].view(np.uint32), alone.view(np.uint32)), len(batch)
    want = oraclelib.opus_file_pcm(oraclelib.opus_decode_file(stereo))
    assert np.abs(alone - want).max() <= 1 / 32767 + 1.2e-7 and (alone != want).sum() <= 2


@pytest.mark.numeric_tolerance
def test_an_empty_sequence_restores_the_stereo_walk(gpu):
    """afg.h: sequences 2p and 2p + 1 are walked together when they are a stream's two channels.  Device level, floats before
    the int16 conversion: a stereo stream behind a mono one and an EMPTY sequence gets the bits it gets alone; without the
    empty sequence it is walked one channel at a time -- other bits, the same tolerance."""
    import torch
    from oraclelib import opus_channel_records
    rng = np.random.default_rng(5)
    pm = afgpu.opus_parse(ob.random_celt_file(rng, 1, 9, pcm_rms=0.05)[0])
    ps = afgpu.opus_parse(ob.random_celt_file(rng, 2, 12, pcm_rms=0.05)[0])

    def run(parts, pad_before=()):
        bases, recs_all, coefs, spans, co, oo, nrec = [0], [], [], [], 0, 0, 0
        for k, p in enumerate(parts):
            base, recs = opus_channel_records(p)
            recs = recs.copy(); recs["coef_off"] += np.uint64(co); recs["out_off"] += np.uint64(oo)
            if k in pad_before:
                bases.append(nrec)                                    # rec_base[k] == rec_base[k+1]: no records
            bases += [nrec + int(b) for b in base[1:]]
            recs_all.append(recs); coefs.append(p["coeffs"]); spans.append((oo, p["pcm_frames"] * p["channels"]))
            nrec += len(recs); co += len(p["coeffs"]); oo += p["pcm_frames"] * p["channels"]
        rec_base = np.array(bases, np.uint64)
        d_out = torch.full((oo,), float("nan"), dtype=torch.float32, device=gpu)
        afgpu.celt_transform(len(rec_base) - 1, torch.from_numpy(rec_base.view(np.int64)).to(gpu),
                             torch.from_numpy(np.concatenate(recs_all).view(np.uint8).copy()).to(gpu),
                             torch.from_numpy(np.concatenate(coefs)).to(gpu), d_out)
        torch.cuda.synchronize()
        o = d_out.cpu().numpy()
        assert not np.isnan(o).any()
        return [o[a:a + n] for a, n in spans]

    alone = run([ps])[0]
    padded = run([pm, ps], pad_before=(1,))
    split = run([pm, ps])
    assert np.array_equal(padded[1].view(np.uint32), alone.view(np.uint32))
    assert np.array_equal(padded[0].view(np.uint32), split[0].view(np.uint32))           # the mono stream does not care
    differ = int((split[1].view(np.uint32) != alone.view(np.uint32)).sum())
    assert differ > 0, "the one-channel walk gave the paired walk's bits: nothing was tested"
    assert float(np.sqrt(np.mean((split[1].astype(np.float64) - alone) ** 2))) <= 1e-6          # far inside the tolerance
