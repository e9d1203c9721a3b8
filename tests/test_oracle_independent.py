"""The oracle against a decoder that was NOT written in this repository (tests/golden/independent_webaudio.npz, made by
tests/golden/make_independent.py in the build container: the WebAudio decodeAudioData of the Chromium inside the image's
`kaleido` package -- Chromium's FFmpeg build for MP3 / Vorbis / FLAC, libopus for Opus).

It is not the reference (SURVEY 8c's pin needs the D decoders or vectors they made: the oracle's header still says "parity
unpinned"), but it is evidence from other hands:
  * MP3: minimp3's arithmetic (the oracle's restatement) and FFmpeg's mp3float are two float implementations of one
    standard-defined decoder -- they must agree to float accuracy on every sample of the real file, with the same delay
    / padding bookkeeping (same length, lag 0);
  * Vorbis: stb_vorbis (restated) and FFmpeg's vorbis decoder, likewise;
  * FLAC: bit-exact integers;
  * Opus: the reference's decoder is a port of FFmpeg's native CELT decoder, the independent one is libopus, the codec's
    normative implementation.  On the two committed files (60 fullband 20 ms frames of random payloads each) they agree
    to 0.5 % RMS (stereo) and 6 % RMS (mono, a signal 80 dB below full scale whose silent stretches libopus flushes to
    exact zeros).  That is gross agreement, not float accuracy: a slip anywhere in the range decoder, the band energies,
    the bit allocation or the PVQ shapes turns the output into unrelated noise (100 % and more).  It is NOT uniform: on
    other random files (other seeds, mixed frame sizes or bandwidths) single frames differ by tens of percent between the
    two decoder lineages -- FFmpeg's CELT decoder is known not to track libopus on everything a random bitstream
    exercises -- and without the reference's own output those differences cannot be assigned to either side.  The Opus
    front-end therefore stays "parity unpinned" in DESIGN.md.
  * Opus, ENCODER-MADE (round 5): three clips that Chromium's MediaRecorder encoded (WebRTC's libopus encoder; every
    packet CELT-only fullband, three 20 ms frames each) and libopus decoded.  Valid streams, which the random payloads
    above are not: the oracle's decode of the same bytes agrees to 2.2e-5 of the signal on the two stationary clips --
    the level of the reference's int16 output conversion -- and to 1.0e-3 on the transient clip, where every bit of the
    difference sits inside short-block frames (0.1-0.6 % of those frames) and their successors' overlap.  That bounds the
    known distance between the FFmpeg CELT lineage (the reference's) and libopus on real material; it still does not say
    which side the reference's own output would be on in those frames."""
import os

import numpy as np
import pytest

import oraclelib

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def vectors():
    return np.load(os.path.join(GOLD, "independent_webaudio.npz"))


def rms(x):
    return float(np.sqrt(np.mean(np.asarray(x, np.float64) ** 2)))


def test_mp3_file_agrees_with_ffmpeg():
    data = open(os.path.join(GOLD, "mathjax_invalid_keypress.mp3"), "rb").read()
    want = vectors()["mp3_pcm"]
    got = oraclelib.mp3_decode_file(data)["pcm"].reshape(-1, 2)
    assert got.shape == want.shape == (23087, 2)                     # same delay / padding trim, sample for sample
    d = got.astype(np.float64) - want
    assert rms(want) > 0.1
    assert rms(d) <= 1e-5 and np.abs(d).max() <= 6e-5, (rms(d), np.abs(d).max())
    # not a coincidence of levels: one sample of misalignment is three orders of magnitude worse
    assert rms(got[1:].astype(np.float64) - want[:-1]) > 100 * rms(d)


def test_vorbis_file_agrees_with_ffmpeg():
    data = open(os.path.join(GOLD, "mathjax_invalid_keypress.ogg"), "rb").read()
    want = vectors()["ogg_pcm"]                                       # WebAudio does not apply the last page's granule trim
    got = oraclelib.vorbis_file_pcm(oraclelib.vorbis_decode_file(data))
    assert got.shape == (22050, 2) and want.shape[0] >= 22050
    d = got.astype(np.float64) - want[:22050]
    assert rms(d) <= 1e-7 and np.abs(d).max() <= 1e-6, (rms(d), np.abs(d).max())


def test_flac_file_agrees_with_ffmpeg():
    import afgpu
    v = vectors()
    data = v["flac_file"].tobytes()
    info, frames, subframes, res = afgpu.flac_parse(data)             # (host parser; the restore below is the oracle's)
    got = oraclelib.flac_transform(frames, subframes, res, info["out_samples"]).reshape(-1, 2) >> 16
    assert np.array_equal(got, v["flac_source"])                      # lossless: the samples that were encoded
    web = v["flac_pcm"].astype(np.float64)
    # WebAudio hands FLAC over as int16 / 32768 for negative and / 32767 for positive values (its sample-format bridge)
    back = np.where(web < 0, web * 32768.0, web * 32767.0)
    assert np.abs(back - got).max() < 0.51 and np.array_equal(np.rint(back).astype(np.int64), got)


def opus_case(tag, channels):
    v = vectors()
    rec = oraclelib.opus_decode_file(v[tag + "_file_gain0"].tobytes())
    assert not isinstance(rec, int) and rec["channels"] == channels and len(rec["frames"]) == 60
    base, recs = oraclelib.opus_channel_records(rec)
    pcm = oraclelib.celt_transform(base, recs, rec["coeffs"], rec["pcm_frames"] * channels).reshape(-1, channels)
    want = v[tag + "_pcm_m78dB"].astype(np.float64)                   # libopus: pre-skip dropped, header gain applied
    got = pcm[312:].astype(np.float64) * 10.0 ** (-20000 / 256.0 / 20.0)
    n = min(len(got), len(want))
    assert n >= 57000
    return got[:n], want[:n]


def test_opus_stereo_agrees_with_libopus():
    got, want = opus_case("opus_stereo", 2)
    assert rms(got - want) <= 0.02 * rms(want), rms(got - want) / rms(want)
    assert rms(got[1:] - want[:-1]) > 20 * rms(got - want)            # aligned to the sample


def test_opus_mono_agrees_with_libopus():
    got, want = opus_case("opus_mono", 1)
    assert rms(got - want) <= 0.10 * rms(want), rms(got - want) / rms(want)
    assert rms(got[1:] - want[:-1]) > 5 * rms(got - want)


@pytest.mark.parametrize("k,channels,bound", [(0, 2, 1e-4), (1, 2, 2e-3), (2, 1, 1e-4)])
def test_encoder_made_opus_agrees_with_libopus(k, channels, bound):
    v = vectors()
    data = v[f"opus_enc{k}_file"].tobytes()
    want = v[f"opus_enc{k}_pcm"].astype(np.float64)
    rec = oraclelib.opus_decode_file(data)
    assert not isinstance(rec, int) and not rec["error"] and rec["channels"] == channels and len(rec["frames"]) == 75
    got = oraclelib.opus_file_pcm(rec).astype(np.float64)
    assert got.shape == want.shape == (72000, channels) and 0.1 < rms(want) < 0.3          # an encoder's level
    d = got - want
    assert rms(d) <= bound * rms(want), rms(d) / rms(want)
    assert rms(got[1:] - want[:-1]) > 100 * rms(d)                                          # aligned to the sample
    # frame by frame: long-block frames that do not follow a transient agree to the int16 conversion's level
    per = np.sqrt((d.reshape(75, 960, channels) ** 2).mean(axis=(1, 2))) / np.sqrt((want.reshape(75, 960, channels) ** 2).mean(axis=(1, 2)))
    short = rec["frames"]["blocks"][:75] > 1
    calm = ~short & ~np.roll(short, 1)
    calm[0] = False
    assert calm.sum() >= 40 and per[calm].max() <= 1e-4, per[calm].max()
    assert per.max() <= 1e-2


# ------------------------------------------------------------------------------------------------------------------
# Round 5: FFmpeg on files from this repository's own bitstream writers (tests/golden/make_independent.py:
# generated_streams).  Random code words, but legal streams -- far more of the front-ends than the two earcons exercise.
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k", range(6))
def test_generated_mp3_agrees_with_ffmpeg(k):
    """MPEG-1 / 2 / 2.5 Layer III at three sample-rate rows, stereo / M-S / mono, every Huffman table with linbits escapes,
    scfsi, preflag, scalefac_scale, subblock gains, long / start / short / stop blocks in an encoder's order, the bit
    reservoir: the oracle's decode (minimp3 restated) equals FFmpeg's to 1.25e-5 rms ABSOLUTE with a largest difference of
    1 / 32768 + rounding -- Chromium hands MP3 over through int16, so this is the resolution of the comparison (the real
    file above: 8.9e-6).  Same length, lag 0.
    Left out of the committed files because the two decoder families read them differently, the reference following
    minimp3: MIXED blocks (5-80 % apart on every granule that is one: FFmpeg's long-standing handling of the switch point)
    and INTENSITY stereo as this writer produces it (its right channel is not empty above the intensity bound, as an
    encoder's is; the decoders place the bound differently on such input)."""
    v = vectors()
    data = v[f"gen_mp3_{k}_file"].tobytes()
    want = v[f"gen_mp3_{k}_pcm"].astype(np.float64)
    rec = oraclelib.mp3_decode_file(data)
    got = rec["pcm"].reshape(-1, rec["channels"]).astype(np.float64)
    assert got.shape == want.shape
    d = got - want
    assert 0.03 < rms(want) < 0.08                                       # an encoder's level
    assert rms(d) <= 1.5e-5 and np.abs(d).max() <= 4e-5, (rms(d), np.abs(d).max())
    assert rms(got[1:] - want[:-1]) > 50 * rms(d)                        # aligned to the sample
    blocks = len(got) // 576
    per = [rms(d[b * 576:(b + 1) * 576]) for b in range(blocks)]
    assert max(per) <= 2e-5                                               # no granule stands out


def test_generated_vorbis_agrees_with_ffmpeg():
    """Six generated Ogg Vorbis files (random set-ups: ordered / dense / sparse code books of lookup type 1, residue types 1
    and 2 with multi-stage cascades, channel coupling, two floors, two mappings, four modes, short and long blocks; packets
    long enough never to end early): the oracle (stb_vorbis restated) equals FFmpeg's decoder to 4e-7 of the signal.
    What the selection means: of the files FFmpeg decodes at all, a quarter to a third agree like this (`gen_ogg_selection`
    = tried, decoded by FFmpeg, agreeing); the others differ in LEVEL from their first packets on -- the packets are random
    words, which an encoder's are not.  Switching generator features off moves the share without explaining it (residue ends
    inside the shortest block: 8-11 of 20 agree; everything optional off, long blocks only: 14 of 20); not isolated further.  Known and left out:
    residue type 0 (the reference inherits stb_vorbis' `n - offset - k` length, which goes negative after the first
    partition: stb_vorbis2.d:1571), sequence_p books (the running sum is not reset per entry when a lookup-1 book is
    expanded, and applied again on decode: :2960-2975, :1351-1358), lookup type 2 (FFmpeg refuses it), packets that end
    early (stb_vorbis stops, FFmpeg reads zeros)."""
    v = vectors()
    tried, decoded, kept = (int(x) for x in v["gen_ogg_selection"])
    assert kept == 6 and decoded >= kept
    for k in range(kept):
        data = v[f"gen_ogg_{k}_file"].tobytes()
        want = v[f"gen_ogg_{k}_pcm"].astype(np.float64)
        got = oraclelib.vorbis_file_pcm(oraclelib.vorbis_decode_file(data)).astype(np.float64)
        n = min(len(got), len(want))
        assert n >= 1024 and len(want) >= len(got) and 0.02 < rms(want[:n]) < 0.1
        assert rms(got[:n] - want[:n]) <= 1e-6 * rms(want[:n]), (k, rms(got[:n] - want[:n]) / rms(want[:n]))
