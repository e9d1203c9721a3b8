"""Ogg Opus host front-end, CELT-only (no device): the product parser (afg_opus_parse) against the oracle restatement of
the reference's decoder (oracle/opus_frontend.c) on generated streams -- random payloads behind every framing code,
frame size, bandwidth and channel combination (tests/opus_bitstream.py) -- and the container rules of opusOpen.

Reference behaviour followed: dopus.d:7791-7829 (headers), :8120-8193 (open), :8011-8059 + :1311-1316 (gain),
:1081-1258 (packet framing), :809-1034 (range decoder), :2128-3678 (CELT frame up to the transform seam)."""
import numpy as np
import pytest

import afgpu
import opus_bitstream as ob
import oraclelib


def same_records(data):
    want = oraclelib.opus_decode_file(data)
    if isinstance(want, int):
        with pytest.raises(afgpu.AfgError) as e:
            afgpu.opus_parse(data)
        assert ("SILK" in str(e.value)) == (want == -3)
        return None, want
    got = afgpu.opus_parse(data)
    for k in ("channels", "preskip", "gain_i", "gain", "error", "declared_frames", "pcm_frames"):
        assert got[k] == want[k], k
    assert got["frames"].tobytes() == want["frames"].tobytes()
    assert got["coeffs"].shape == want["coeffs"].shape
    assert np.array_equal(got["coeffs"].view(np.uint32), want["coeffs"].view(np.uint32))      # bit-exact coefficients
    return got, want


@pytest.mark.parametrize("seed", range(12))
def test_random_streams_bit_exact(seed):
    rng = np.random.default_rng(1000 + seed)
    frames = 0
    for _ in range(12):
        ch = int(rng.integers(1, 3))
        data, pkts = ob.random_celt_file(rng, ch, int(rng.integers(1, 60)), preskip=int(rng.integers(0, 121)),
                                         gain=int(rng.integers(-3000, 3000)) if rng.random() < 0.5 else 0)
        got, want = same_records(data)
        assert got is not None and not got["error"]
        assert got["pcm_frames"] == sum(c * fs for c, fs in map(ob.packet_frames, pkts))
        frames += len(got["frames"])
    assert frames > 100


@pytest.mark.parametrize("config", range(16, 32))
@pytest.mark.parametrize("channels,stereo", [(1, False), (1, True), (2, False), (2, True)])
def test_every_celt_configuration(config, channels, stereo):
    """all four bandwidths x four frame sizes, coded mono / stereo into mono / stereo output (down-mix, duplication)"""
    rng = np.random.default_rng(config * 8 + channels * 2 + stereo)
    pkts = [ob.packet(rng, config, stereo, 0, sizes=[int(s)]) for s in rng.integers(0, 400, 25)]
    pkts += [ob.packet(rng, config, stereo, 0, sizes=[1275]), ob.packet(rng, config, stereo, 0, sizes=[0])]
    first = ob.packet_frames(pkts[0])[1]
    got, _ = same_records(ob.ogg_opus(pkts, channels, preskip=min(312, first), rng=rng))
    assert len(got["frames"]) == 27
    assert set(got["frames"]["frame_size"]) == {ob.CELT_FRAME_SIZES[config & 3]}
    scale = set(got["frames"]["imdct_scale"])
    assert scale == ({0.5} if stereo and channels == 1 else {1.0})            # dopus.d:3663-3666
    assert set(got["frames"]["blocks"]) <= {1, 1 << (config & 3)}              # long block, or 120-sample short blocks (dopus.d:3630)
    assert (got["frames"]["pf_period_new"][got["frames"]["pf_gains_new"][:, 0] != 0] >= 15).all()


def test_framing_codes_and_padding():
    rng = np.random.default_rng(5)
    pkts = [ob.packet(rng, 27, True, 1, sizes=[40]), ob.packet(rng, 27, True, 2, sizes=[30, 55]), ob.packet(rng, 27, True, 2, sizes=[252, 3]),
            ob.packet(rng, 26, False, 3, sizes=[20], count=4, vbr=False), ob.packet(rng, 26, False, 3, sizes=[10, 300, 0, 25], count=4, vbr=True),
            ob.packet(rng, 25, True, 3, sizes=[33], count=2, vbr=False, pad=1), ob.packet(rng, 25, True, 3, sizes=[33], count=2, vbr=False, pad=255),
            ob.packet(rng, 24, True, 3, sizes=[12, 13, 14], count=3, vbr=True, pad=600)]
    got, _ = same_records(ob.ogg_opus(pkts, 2, preskip=0, packets_per_page=3))
    assert list(got["frames"]["frame_size"]) == [960] * 6 + [480] * 8 + [240] * 4 + [120] * 3
    assert not got["error"]


def test_bad_packet_ends_the_records_with_the_error_flag():
    rng = np.random.default_rng(6)
    good = [ob.packet(rng, 31, True, 0, sizes=[100]) for _ in range(5)]
    for bad in (ob.toc(31, True, 1) + bytes(7),                       # code 1 with an odd payload
                ob.toc(31, True, 3),                                  # code 3 without its count byte
                ob.toc(31, True, 3) + bytes([0]),                     # zero frames
                ob.toc(31, True, 3) + bytes([49]) + bytes(49),        # 49 frames
                ob.toc(31, True, 3) + bytes([4]) + bytes(16),         # 4 x 20 ms x ... more than 60 ms of stereo: 3840 samples
                ob.toc(31, True, 3) + bytes([0x41, 200]) + bytes(20),  # padding longer than the packet
                ob.toc(31, True, 2) + bytes([250]) + bytes(20)):      # first frame longer than the packet
        got, want = same_records(ob.ogg_opus(good[:3] + [bad] + good[3:], 2, preskip=0))
        assert got["error"] and len(got["frames"]) == 3


def test_mono_stream_takes_120_ms_packets_stereo_does_not():
    rng = np.random.default_rng(7)
    long_pkt = ob.packet(rng, 31, False, 3, sizes=[50], count=6, vbr=False)           # 6 x 20 ms
    got, _ = same_records(ob.ogg_opus([long_pkt], 1, preskip=0))
    assert not got["error"] and got["pcm_frames"] == 5760
    got, _ = same_records(ob.ogg_opus([long_pkt], 2, preskip=0))
    assert got["error"] and got["pcm_frames"] == 0                                    # the reference's frame buffer holds 60 ms per channel


def test_gain_is_header_plus_r128_tag_with_the_unsigned_header_read():
    rng = np.random.default_rng(8)
    pkts = [ob.packet(rng, 30, True, 0) for _ in range(3)]
    cases = [(0, (), 0), (256, (), 256), (-256, (), 32767),                          # a negative header gain reads as 65280 and clamps (:516, :1311)
             (100, (b"R128_TRACK_GAIN=-356",), -256), (0, (b"r128_track_gain=+77",), 77), (0, (b"  R128_TRACK_GAIN=12  ",), 12),
             (0, (b"ARTIST=x", b"R128_TRACK_GAIN=-32768"), -32768), (0, (b"R128_TRACK_GAIN=32768",), 0), (0, (b"R128_TRACK_GAIN=1x",), 0),
             (0, (b"R128_TRACK_GAIN=",), 0), (0, (b"R128_ALBUM_GAIN=55",), 0), (-1, (b"R128_TRACK_GAIN=-32768",), 32767)]
    for header, comments, expect in cases:
        got, _ = same_records(ob.ogg_opus(pkts, 2, preskip=0, gain=header, comments=comments))
        assert got["gain_i"] == expect, (header, comments)
        if expect:
            assert abs(got["gain"] - 10 ** (expect / 5120)) < 1e-6 * got["gain"]
        else:
            assert got["gain"] == 1.0


def test_open_rules():
    rng = np.random.default_rng(9)
    pkts = [ob.packet(rng, 31, False, 0, sizes=[60]) for _ in range(4)]

    def opens(data):
        got, want = same_records(data)
        return got is not None

    assert opens(ob.ogg_opus(pkts, 1, preskip=960))
    assert not opens(ob.ogg_opus(pkts, 1, preskip=960, bos=False))                              # the head packet sits on a BOS page
    assert not opens(ob.ogg_opus(pkts, 1, head=ob.opus_head(1, 960, version=0x11)))             # version nibble
    assert opens(ob.ogg_opus(pkts, 1, head=ob.opus_head(1, 960, version=0x0f)))
    assert not opens(ob.ogg_opus(pkts, 1, head=ob.opus_head(1, 960)[:18]))                      # shorter than 19 bytes
    assert not opens(ob.ogg_opus(pkts, 3, preskip=0))                                           # mapping 0 is mono / stereo
    assert not opens(ob.ogg_opus(pkts, 0, preskip=0))
    assert not opens(ob.ogg_opus(pkts, 2, head=ob.opus_head(2, 0, map_type=1, extra=bytes([1, 1, 0, 1]))))   # mapping families: refused here
    assert not opens(ob.ogg_opus(pkts, 1, tags=b"OpusTagz" + bytes(8)))
    assert opens(ob.ogg_opus(pkts, 1, preskip=0, tags=b"OpusTags"))                             # no comment block at all
    assert not opens(ob.ogg_opus([], 1, preskip=0))                                             # nothing behind the tags
    assert not opens(ob.ogg_opus(pkts, 1, preskip=4000))                                        # last granule < pre-skip
    assert not opens(ob.ogg_opus(pkts, 1, preskip=100, first_granule=99, packets_per_page=1))   # first audio page's granule < pre-skip
    assert opens(ob.ogg_opus(pkts, 1, preskip=100, first_granule=100, packets_per_page=1))
    assert not opens(b"OggS" + bytes(200))
    assert not opens(open(__file__, "rb").read())


def test_length_is_the_last_granule_minus_preskip_and_preskip_is_not_dropped():
    rng = np.random.default_rng(10)
    pkts = [ob.packet(rng, 31, True, 0, sizes=[80]) for _ in range(10)]
    got, _ = same_records(ob.ogg_opus(pkts, 2, preskip=312, trim=500))
    assert got["pcm_frames"] == 9600 and got["declared_frames"] == 9600 - 500 - 312
    assert got["frames"]["out_off"][0] == 0 and got["frames"]["out_stride"][0] == 2
    assert list(got["frames"]["out_off"]) == [960 * 2 * k for k in range(10)]
    assert list(got["frames"]["coef_off"]) == [960 * 2 * k for k in range(10)]


def test_silk_and_hybrid_files_are_refused_as_a_whole():
    rng = np.random.default_rng(11)
    celt = [ob.packet(rng, 31, True, 0, sizes=[80]) for _ in range(4)]
    for config in (0, 11, 12, 15):
        other = ob.toc(config, True, 0) + rng.bytes(40)
        got, want = same_records(ob.ogg_opus(celt + [other] + celt, 2, preskip=0))
        assert got is None and want == -3


def test_a_page_with_a_bad_checksum_ends_the_stream():
    rng = np.random.default_rng(12)
    pkts = [ob.packet(rng, 31, True, 0, sizes=[80]) for _ in range(12)]
    whole, _ = same_records(ob.ogg_opus(pkts, 2, preskip=0, packets_per_page=3))
    cut, _ = same_records(ob.ogg_opus(pkts, 2, preskip=0, packets_per_page=3, corrupt_page=4))   # pages: head, tags, 4 audio
    assert whole["pcm_frames"] == 12 * 960 and cut["pcm_frames"] == 6 * 960
    assert cut["declared_frames"] == 6 * 960                    # the last *valid* page in sequence
    assert np.array_equal(cut["coeffs"], whole["coeffs"][:len(cut["coeffs"])])


def test_tags_over_several_pages_and_truncated_files():
    rng = np.random.default_rng(13)
    pkts = [ob.packet(rng, 29, True, 0, sizes=[80]) for _ in range(8)]
    big = ob.opus_tags(comments=[b"COMMENT=" + bytes(rng.integers(65, 91, 70000).astype(np.uint8)), b"R128_TRACK_GAIN=-300"])
    data = ob.ogg_opus(pkts, 2, preskip=0, tags=big)
    got, _ = same_records(data)
    assert got["gain_i"] == -300 and got["pcm_frames"] == 8 * 240
    for cut in (len(data) - 1, len(data) - 200, len(data) // 2, 100, 47, 46, 0):
        same_records(data[:cut])


def test_energy_memory_and_noise_seed_carry_across_frames():
    """the same packet decodes differently after different predecessors (inter-frame energy prediction, anti-collapse
    history, the LCG seed taken from the previous frame's final range)"""
    rng = np.random.default_rng(14)
    a, b, c = (ob.packet(rng, 31, True, 0, sizes=[90]) for _ in range(3))
    x, _ = same_records(ob.ogg_opus([a, c], 2, preskip=0))
    y, _ = same_records(ob.ogg_opus([b, c], 2, preskip=0))
    assert not np.array_equal(x["coeffs"][1920:], y["coeffs"][1920:])
    z, _ = same_records(ob.ogg_opus([a, c, a, c], 2, preskip=0))
    assert np.array_equal(z["coeffs"][:3840], x["coeffs"])


def test_statistics_of_what_the_random_payloads_reach():
    """the generator is only as good as the decoder paths it reaches: over a few hundred frames the post-filter, silence,
    transients of every size, mono-in-stereo and stereo-in-mono all occur"""
    rng = np.random.default_rng(15)
    seen_pf = seen_tr = 0
    n = 0
    for ch in (1, 2):
        data, _ = ob.random_celt_file(rng, ch, 150, preskip=0)
        got = afgpu.opus_parse(data)
        fr = got["frames"]
        n += len(fr)
        seen_pf += int((fr["pf_gains_new"][:, 0] != 0).sum())
        seen_tr += int((fr["blocks"] > 1).sum())
        assert set(np.unique(fr["blocks"])) >= {1, 2, 4, 8}
        assert np.isfinite(got["coeffs"]).all()
    assert seen_pf > n // 4 and seen_tr > n // 20
