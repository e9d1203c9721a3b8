"""Ogg Vorbis host front-end (no device): the product parser (afg_vorbis_parse) against the oracle restatement of
stb_vorbis on a real file and on damaged variants, and the real file against the MP3 encoding of the same sound.

Reference behaviour followed: stb_vorbis2.d:984-1152 (pages, packets), :2669-3266 (setup), :2300-2597 (packet decode
and sample bookkeeping), :2606-2657 (finish_frame), :3797-3868 (stream length)."""
import os

import numpy as np
import pytest

import afgpu
import oraclelib

HERE = os.path.dirname(__file__)
OGG = os.path.join(HERE, "golden", "mathjax_invalid_keypress.ogg")
MP3 = os.path.join(HERE, "golden", "mathjax_invalid_keypress.mp3")


def overlap_lengths(fl, bs0, bs1):
    """(left window length, right window length) of a packet with these flags (stb_vorbis2.d:2333-2349)"""
    n = bs1 if fl & 1 else bs0
    left = bs0 // 2 if (fl & 1) and not (fl & 2) else n // 2
    right = bs0 // 2 if (fl & 1) and not (fl & 4) else n // 2
    return left, right


def cut_at_window_break(want):
    """the product ends a stream where a block's left window does not match its predecessor's right window (the
    reference overlaps the unequal windows and goes on): trim the oracle's records to that point"""
    fl, bs0, bs1, ch = want["pflags"], want["blocksize0"], want["blocksize1"], want["channels"]
    for p in range(1, len(fl)):
        if overlap_lengths(fl[p - 1], bs0, bs1)[1] != overlap_lengths(fl[p], bs0, bs1)[0]:
            spec_n = sum((bs1 if f & 1 else bs0) // 2 * ch for f in fl[:p])
            out = dict(want)
            for k in ("pflags", "take_from", "take_count"):
                out[k] = want[k][:p]
            out["spec"] = want["spec"][:spec_n]
            out["pcm_frames"] = int(out["take_count"].sum())
            return out
    return want


def same_records(data, upstream_seek=False):
    want = oraclelib.vorbis_decode_file(data, seek_clears_eof=upstream_seek)
    if want is None:
        with pytest.raises(afgpu.AfgError):
            afgpu.vorbis_parse(data)
        return None, None
    want = cut_at_window_break(want)
    got = afgpu.vorbis_parse(data)
    for k in ("channels", "sample_rate", "blocksize0", "blocksize1", "total_samples", "pcm_frames"):
        assert got[k] == want[k], k
    np.testing.assert_array_equal(got["pflags"] & 7, want["pflags"], err_msg="pflags")
    for k in ("take_from", "take_count"):
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    assert got["spec"].shape == want["spec"].shape
    assert np.array_equal(got["spec"].view(np.uint32), want["spec"].view(np.uint32))      # bit-exact spectra
    declared_tail_is_zero(got["pflags"], want)
    return got, want


def declared_tail_is_zero(pflags, want):
    """bits 4..7 of a long packet's flags (afg.h AFG_VORBIS_NZ_EIGHTHS) are the product's own: every long packet carries a
    declaration, no short one does, and in the ORACLE's spectra the declared-empty eighths hold +0.0 and nothing else"""
    ch, at = want["channels"], 0
    for fl in pflags:
        n2 = (want["blocksize1"] if fl & 1 else want["blocksize0"]) // 2
        if fl & 1:
            e = (int(fl) >> 4) - 1
            assert 0 <= e <= 8
            for c in range(ch):
                tail = want["spec"][at + c * n2 + e * (n2 // 8):at + (c + 1) * n2]
                assert not tail.view(np.uint32).any(), "a declared-empty eighth holds a nonzero (or -0.0)"
        else:
            assert fl >> 4 == 0
        at += n2 * ch


def test_real_file():
    got, want = same_records(open(OGG, "rb").read())
    assert (got["channels"], got["sample_rate"], got["blocksize0"], got["blocksize1"]) == (2, 44100, 256, 2048)
    assert got["total_samples"] == 22050 == got["pcm_frames"]          # the last page's granule position truncates the tail
    assert got["take_count"][0] == 0                                     # the first frame only primes the overlap (:2659)
    assert set(got["pflags"] & 7) >= {0, 7}                              # short and long blocks
    pcm = oraclelib.vorbis_file_pcm(got)
    assert pcm.shape == (22050, 2) and np.isfinite(pcm).all() and 0.3 < np.abs(pcm).max() < 1.0


def test_two_codecs_agree_on_the_sound():
    """The same earcon ships as MP3 and as Ogg Vorbis: two unrelated bitstreams, two unrelated front-ends, two
    transform oracles -- the decoded waveforms must be the same sound (what no single-codec self-check can show)."""
    v = oraclelib.vorbis_file_pcm(oraclelib.vorbis_decode_file(open(OGG, "rb").read()))
    m = oraclelib.mp3_decode_file(open(MP3, "rb").read())["pcm"].reshape(-1, 2)
    n = min(len(v), len(m))
    for c in range(2):
        a, b = v[:n, c].astype(np.float64), m[:n, c].astype(np.float64)
        assert np.corrcoef(a, b)[0, 1] > 0.9999                          # sample-aligned (both trim their codec delay)
        assert np.sqrt(np.mean((a - b) ** 2)) < 0.1 * np.sqrt(np.mean(b ** 2))


def test_truncation_and_damage_agree_with_the_oracle():
    d = open(OGG, "rb").read()
    audio_page = 58 + 27 + 17 + 3877                                     # start of the third page
    # A cut file has no page with the last-page flag: the stream-length scan runs into the end of the data.  The
    # reference's seek (unlike upstream stb_vorbis') does not reset the eof flag afterwards, so its decoder stops
    # after the frame it primed at open time: nothing is delivered.  The product keeps decoding what is there, like
    # upstream; both behaviours are pinned here.
    for cut in (len(d) - 1, len(d) - 200, audio_page + 27 + 26 + 700, audio_page + 30, audio_page, audio_page - 5, 3000, 58, 40):
        same_records(d[:cut], upstream_seek=True)
        ref = oraclelib.vorbis_decode_file(d[:cut])
        assert ref is None or ref["pcm_frames"] == 0
    rng = np.random.default_rng(3)
    body0 = audio_page + 27 + 26
    agree = 0
    for trial in range(40):
        v = bytearray(d)
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(body0, len(v)))
            v[pos] ^= 1 << int(rng.integers(0, 8))                       # audio payload damage (page CRCs are not checked
        same_records(bytes(v), upstream_seek=True)                       # while decoding, :1019); the length scan does
        agree += 1                                                       # check them, fails, and trips the same eof quirk
    assert agree == 40


def test_not_vorbis():
    d = open(OGG, "rb").read()
    for blob in (b"", b"OggS" + bytes(100), d[:58] + bytes(200), bytes(5000), open(MP3, "rb").read(),
                 d[:28] + b"\x01OpusHead" + d[37:]):
        with pytest.raises(afgpu.AfgError):
            afgpu.vorbis_parse(blob)
        assert oraclelib.vorbis_decode_file(blob) is None


@pytest.mark.parametrize("channels,bs", [(1, (256, 2048)), (2, (256, 2048)), (2, (512, 512)), (3, (256, 1024)), (6, (1024, 4096)),
                                         (2, (2048, 8192)), (16, (256, 256))])
def test_synthetic_streams(channels, bs):
    """Random legal set-ups and random payloads (tests/vorbis_bitstream.py): residue types 0/1/2, lookup 1 and 2 books,
    sequence_p, sparse / ordered length lists, several submaps, coupling, 1..16 channels, every block-size pairing."""
    import vorbis_bitstream as vb
    decoded = 0
    for seed in range(6):
        data = vb.make_file(100 * channels + seed, channels=channels, bs=bs, n_packets=20,
                            residue_types=[(0, 1), (1, 2), (2, 0), (2, 2), (0, 0), (1, 1)][seed])
        got, want = same_records(data)
        assert got is not None, "generated stream rejected"
        assert got["channels"] == channels and (got["blocksize0"], got["blocksize1"]) == bs
        decoded += len(got["pflags"])
        assert np.isfinite(got["spec"]).all()
    assert decoded >= 6 * 10


def test_inconsistent_window_flags_end_the_stream():
    """prev/next flags that contradict the neighbouring block size: the reference overlaps windows of different
    lengths and carries on; the device transform needs equal lengths, so the product ends the stream there."""
    import vorbis_bitstream as vb
    hit = 0
    for seed in range(12):
        data = vb.make_file(4000 + seed, channels=2, bs=(256, 2048), n_packets=14, break_windows_at=6)
        want = oraclelib.vorbis_decode_file(data)
        got = afgpu.vorbis_parse(data)
        if len(got["pflags"]) == len(want["pflags"]):
            continue                                       # packet 6 was a short block: nothing to break
        hit += 1
        n = len(got["pflags"])
        assert n == 6 and len(want["pflags"]) > n
        np.testing.assert_array_equal(got["pflags"] & 7, want["pflags"][:n])
        assert np.array_equal(got["spec"].view(np.uint32), want["spec"][:len(got["spec"])].view(np.uint32))
    assert hit >= 3


# ---- the tail of the packet decode left to the device: records + residues instead of spectra ----
def same_spectra_through_records(data):
    """afg_vorbis_parse_r -> oracle restatement of coupling / do_floor on the records == afg_vorbis_parse's spectra"""
    full = afgpu.vorbis_parse(data)
    r = afgpu.vorbis_parse_r(data)
    for k in ("channels", "sample_rate", "blocksize0", "blocksize1", "total_samples", "pcm_frames"):
        assert r[k] == full[k], k
    for k in ("pflags", "take_from", "take_count"):
        np.testing.assert_array_equal(r[k], full[k], err_msg=k)
    assert len(r["fl_packets"]) == len(full["pflags"]) and len(r["fl_curves"]) == len(full["pflags"]) * full["channels"]
    assert r["spec"].shape == full["spec"].shape
    got = oraclelib.vorbis_floor(r["fl_packets"], r["fl_curves"], r["fl_points"], r["fl_steps"], r["spec"])
    assert np.array_equal(got.view(np.uint32), full["spec"].view(np.uint32))
    # the records are what the header says: packets tile the plane, curves start at x = 0 with ascending x
    at = 0
    for k, fl in zip(r["fl_packets"], r["pflags"]):
        n2 = (full["blocksize1"] if fl & 1 else full["blocksize0"]) // 2
        assert (int(k["spec_off"]), int(k["n2"]), int(k["channels"])) == (at, n2, full["channels"])
        at += n2 * full["channels"]
    for c in r["fl_curves"]:
        if c["n_points"]:
            x = r["fl_points"][int(c["point_off"]):int(c["point_off"]) + int(c["n_points"]), 0]
            assert x[0] == 0 and (np.diff(x) >= 0).all() and c["n_points"] >= 2
    return r, full


def test_residue_records_of_the_real_file():
    r, full = same_spectra_through_records(open(OGG, "rb").read())
    assert len(r["fl_steps"]) == 2 and all(tuple(s) == (0, 1) for s in r["fl_steps"])   # two mappings, one stereo coupling step each
    assert not np.array_equal(r["spec"].view(np.uint32), full["spec"].view(np.uint32))
    assert (r["fl_curves"]["n_points"] == 0).any()                                   # silent channels occur in the earcon's tail


@pytest.mark.parametrize("channels,bs", [(1, (256, 2048)), (2, (256, 2048)), (2, (512, 512)), (3, (256, 1024)), (6, (1024, 4096)),
                                         (2, (2048, 8192)), (16, (256, 256))])
def test_residue_records_of_synthetic_streams(channels, bs):
    import vorbis_bitstream as vb
    steps = 0
    for seed in range(6):
        data = vb.make_file(100 * channels + seed, channels=channels, bs=bs, n_packets=20,
                            residue_types=[(0, 1), (1, 2), (2, 0), (2, 2), (0, 0), (1, 1)][seed])
        r, _ = same_spectra_through_records(data)
        steps += int(r["fl_packets"]["n_steps"].sum())
    assert channels == 1 or steps > 0                                               # coupled mappings were exercised


def test_residue_records_of_damaged_files():
    d = open(OGG, "rb").read()
    rng = np.random.default_rng(11)
    for trial in range(25):
        v = bytearray(d)
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(4000, len(v)))
            v[pos] ^= 1 << int(rng.integers(0, 8))
        same_spectra_through_records(bytes(v))


def test_a_comment_header_whose_lengths_overrun_the_packet():
    """stb_vorbis2.d:2736-2770: the comment header is read with get8_packet / get32_packet, which answer EOP (-1) past the end
    of the packet -- a vendor length that a flipped bit has made 512 MB runs into the end, the comment count read there is
    negative, the framing "byte" 0xff: the header passes and the stream decodes (tests/golden/soak_r05_vendor_length.ogg, a
    damaged generated file).  What fails in the reference is an allocation of a size that is not positive: a vendor or
    comment length of -1 or less."""
    import struct
    here = os.path.dirname(os.path.abspath(__file__))
    data = open(os.path.join(here, "golden", "soak_r05_vendor_length.ogg"), "rb").read()
    got, want = same_records(data)
    assert want is not None and len(want["pflags"]) == 24 and want["pcm_frames"] > 10000
    # the same file with the vendor length set to -1 and to 0x7fffffff (len + 1 wraps): refused by both
    at = data.index(b"\x03vorbis") + 7
    for bad in (-1, 0x7fffffff, -2000):
        blob = data[:at] + struct.pack("<i", bad) + data[at + 4:]
        assert oraclelib.vorbis_decode_file(blob) is None
        with pytest.raises(afgpu.AfgError):
            afgpu.vorbis_parse(blob)


@pytest.mark.parametrize("name", ["soak_r05_book_miss_sorted_a.ogg", "soak_r05_book_miss_sorted_b.ogg", "soak_r05_book_miss_linear.ogg",
                                  "soak_r05_read_past_packet_end.ogg"])
def test_code_books_whose_lengths_leave_the_tree_incomplete(name):
    """A flipped bit in a code-word length of the setup header: the book still opens (stb_vorbis2.d:691-737 refuses only a
    length list with too many words), and the stream now holds words that are not in it.  The reference reads such a word
    through its sorted list as the nearest listed word below it, taken with that word's length (:1211-1240: books of more
    than 8 entries that list any word -- the `sorted` files), or fails the symbol, drops the bits it has fetched and reads on
    from the next byte (:1242-1262 -- the `linear` file, where the floor decode continues behind the failure).  Damaged
    generated files found by the header-damage soak (tools/soak_damaged.py, AFG_SOAK_HDR=1).  The last file is the other end of
    the fetched-bit count the search needs: a fixed-width read that runs past the end of a packet (valid_bits = INVALID_BITS)
    followed by a symbol decode, which must then fail (:1154-1198)."""
    here = os.path.dirname(os.path.abspath(__file__))
    data = open(os.path.join(here, "golden", name), "rb").read()
    got, want = same_records(data)
    assert want is not None and len(want["pflags"]) == 8
    same_spectra_through_records(data)


def test_the_continued_flag_of_the_comment_headers_page_is_not_looked_at():
    """stb_vorbis2.d:2732: start_decoder opens the second page itself (start_page, then start_packet with a page already
    open), so the check start_packet makes when IT turns a page (:1056-1069 and :1071-1090, "continued packet flag invalid") never sees
    that page's flag.  A generated file with that one bit set decodes in the reference; on any later page the same bit
    ends the stream."""
    here = os.path.dirname(os.path.abspath(__file__))
    data = open(os.path.join(here, "golden", "soak_r05_comment_page_continued.ogg"), "rb").read()
    assert data[58:62] == b"OggS" and data[58 + 5] & 1
    got, want = same_records(data)
    assert want is not None and len(want["pflags"]) == 8


def test_no_stream_length_when_the_page_search_starts_at_the_top_of_the_file():
    """stb_vorbis2.d:3407: `if (retry_loc - 25 > f.stream_len) return 0` is unsigned, so a page search that meets an 'O' in
    the first 24 bytes of the file gives up.  The length scan starts at first_audio_page_offset when the file is shorter
    than 64 KB, and that is 0 when the setup header does not end its page (here: a damaged lacing value cuts it one byte
    short): such a file reports no length (0) although its last page carries one."""
    here = os.path.dirname(os.path.abspath(__file__))
    data = open(os.path.join(here, "golden", "soak_r05_headers_end_inside_a_page.ogg"), "rb").read()
    got, want = same_records(data)
    assert want is not None and want["total_samples"] == 0 and got["total_samples"] == 0
