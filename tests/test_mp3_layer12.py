"""MPEG Layer I / II host front-end (no device): the product parser against the oracle restatement of minimp3's Layer
I / II path, the standard's requantisation formula on hand-written Layer I frames, and the packing of 12-slot synthesis
granules into the transform stage's 18-slot blocks.

Reference behaviour followed: minimp3.d:286-346 (allocation tables by version / bit rate / sampling rate), :348-435
(allocation, scalefactor selection, scalefactors), :437-485 (sample groups, scaling, joint-stereo copy), :1557-1578
(frame loop: Layer II synthesises each third of the frame, Layer I the whole), :1408-1434 (synthesis granule of
`nbands` slots)."""
import numpy as np
import pytest

import afgpu
import mp3_l12_bitstream as lb
import oraclelib

L12 = np.uint32(0x40000000)        # oracle: a 12-slot synthesis granule
SUBBAND = np.uint32(0x80000000)    # product: AFG_MP3_SUBBAND


def slots_of_oracle(want):
    """per run: [n_slots, channels, 32] subband samples in time order"""
    ch, out, at = want["channels"], [], 0
    for g in want["runs"]:
        blocks = want["coef"][at * ch:(at + int(g)) * ch].reshape(int(g), ch, 32, 18)[:, :, :, :12]
        out.append(blocks.transpose(0, 3, 1, 2).reshape(-1, ch, 32))
        at += int(g)
    return out


def slots_of_product(parsed):
    info, runs, coef, flags, _ = parsed
    ch, out, at = info["channels"], [], 0
    for g in runs:
        blocks = coef[at * ch:(at + int(g)) * ch].reshape(int(g), ch, 32, 18)
        out.append(blocks.transpose(0, 3, 1, 2).reshape(-1, ch, 32))
        at += int(g)
    return out


def same_stream(data):
    want = oraclelib.mp3_decode_file(data)
    if want is None:
        with pytest.raises(afgpu.AfgError):
            afgpu.mp3_parse(data)
        return None, None
    parsed = afgpu.mp3_parse(data)
    info, runs, coef, flags, copies = parsed
    assert want["layer"] in (1, 2)
    assert (info["channels"], info["hz"], info["tagged"], info["start_delay"]) == (want["channels"], want["hz"], 0, 0)
    assert info["declared_samples"] == want["declared_samples"]
    assert (want["flags"] == L12).all() and ((flags & SUBBAND) != 0).all()
    assert len(runs) == len(want["runs"])
    for got, ref, g_blocks, g_gran in zip(slots_of_product(parsed), slots_of_oracle(want), runs, want["runs"]):
        assert int(g_blocks) == (12 * int(g_gran) + 17) // 18                    # three granules in two blocks, the tail padded
        assert np.array_equal(got[:len(ref)].view(np.uint32), ref.view(np.uint32))   # bit-exact subband samples
        assert not got[len(ref):].view(np.uint32).any()                           # silent padding slots
    assert info["pcm_samples"] == len(want["pcm"]) == int(copies[:, 1].sum())
    # records -> transform oracle (18 slots at a time) -> copy plan == the reference-shaped drive (12 slots at a time)
    plane = oraclelib.mp3_transform(runs, np.full(len(runs), info["channels"], np.uint8), coef.reshape(-1), flags)
    got = np.concatenate([plane[int(s):int(s + n)] for s, n in copies]) if len(copies) else np.zeros(0, np.float32)
    assert np.array_equal(got.view(np.uint32), want["pcm"].view(np.uint32))
    return parsed, want


# (MPEG-2.5 exists for Layer III only: hdr_valid accepts the 0xFFE sync with layer bits 01 alone, minimp3.d:232-239)
CONFIGS = [(layer, version, sr, mode) for layer in (1, 2) for version in ("mpeg1", "mpeg2") for sr in (0, 1, 2)
           for mode in ("stereo", "joint", "dual", "mono")]


@pytest.mark.parametrize("layer,version,sr,mode", CONFIGS)
def test_random_payloads_bit_exact(layer, version, sr, mode):
    """every allocation-table choice of L12_subband_alloc_table: all bit rates of the configuration in one stream"""
    rng = np.random.default_rng([layer, ("mpeg1", "mpeg2", "mpeg25").index(version), sr, ("stereo", "joint", "dual", "mono").index(mode)])
    data = lb.random_file(rng, layer, 45, version, sr=sr, mode=mode, vary_bitrate=True)
    parsed, want = same_stream(data)
    assert parsed is not None and want["layer"] == layer
    frames = len(want["pcm"]) // want["channels"]
    assert frames > 0 and frames % (384 if layer == 1 else 1152) == 0
    assert np.isfinite(want["pcm"]).all()


@pytest.mark.parametrize("mode,mode_ext", [("mono", 0), ("stereo", 0), ("joint", 0), ("joint", 2), ("joint", 3)])
@pytest.mark.parametrize("version,sr", [("mpeg1", 0), ("mpeg1", 2), ("mpeg2", 1)])
def test_layer1_frames_decode_to_the_standards_requantisation(mode, mode_ext, version, sr):
    rng = np.random.default_rng(7 + mode_ext)
    frames, wants = zip(*[lb.layer1_frame(rng, version, 12, sr, mode, mode_ext) for _ in range(12)])
    parsed, want = same_stream(b"".join(frames))
    assert parsed[0]["channels"] == (1 if mode == "mono" else 2) and len(want["runs"]) == 1
    got = slots_of_oracle(want)[0]                                   # [slots, ch, 32]
    ref = np.concatenate([w.transpose(2, 0, 1) for w in wants])      # [12 per frame, ch, 32]
    assert got.shape == ref.shape
    assert (np.abs(got - ref) <= 5e-6 * np.abs(ref)).all()           # six-digit table constants (relative 1.6e-6) + float arithmetic
    assert np.abs(ref).max() > 1e-3


def test_mpeg25_layer_two_is_not_a_stream():
    rng = np.random.default_rng(2)
    parsed, want = same_stream(lb.random_file(rng, 2, 20, "mpeg25"))
    assert parsed is None and want is None


def test_crc_word_is_skipped():
    rng = np.random.default_rng(3)
    with_crc = lb.random_file(rng, 2, 20, crc=True)
    parsed, want = same_stream(with_crc)
    assert parsed is not None and 0 < len(want["pcm"]) <= 20 * 1152 * 2 and len(want["pcm"]) % 2304 == 0


def test_a_frame_whose_bits_run_out_is_dropped_and_restarts_the_decoder():
    """minimp3.d:1573-1577: pos > limit -> mp3dec_init, the frame yields nothing and the next one starts from zero state.
    Layer II at 32 kbit/s mono with every allocation index at its maximum asks for far more bits than the frame holds."""
    rng = np.random.default_rng(4)
    good = [lb.random_frame(rng, 2, "mpeg1", 2, 0, "mono", padding=0) for _ in range(16)]
    greedy = lb.random_frame(rng, 2, "mpeg1", 2, 0, "mono", padding=0, fill=0xff)
    parsed, want = same_stream(b"".join(good[:8] + [greedy] + good[8:]))
    assert len(want["runs"]) >= 2                                    # the state was reset at least once
    assert parsed[0]["pcm_samples"] < 17 * 1152


def test_layer_change_ends_the_stream_like_the_reference():
    rng = np.random.default_rng(5)
    two = b"".join(lb.random_frame(rng, 2, fill=0) for _ in range(14))       # silent frames: they always fit
    one = b"".join(lb.random_frame(rng, 1, fill=0) for _ in range(14))
    parsed, want = same_stream(two + one)
    # (the last frame before the change has no matching header behind it: the decoder's sync check gives it up, minimp3.d:1500-1507)
    assert want["layer"] == 2 and parsed[0]["pcm_samples"] in (13 * 1152 * 2, 14 * 1152 * 2)
    parsed, want = same_stream(one + two)
    assert want["layer"] == 1 and parsed[0]["pcm_samples"] in (13 * 384 * 2, 14 * 384 * 2)


def test_quantised_upload_declines_layer_two():
    rng = np.random.default_rng(6)
    with pytest.raises(afgpu.AfgError):
        afgpu.mp3_parse_q(lb.random_file(rng, 2, 12))


def test_chunk_alignment_of_layer_one():
    """a Layer I frame is 12 slots: 3 frames fill 2 blocks.  Whatever the frame count, blocks = ceil(12 n / 18)."""
    rng = np.random.default_rng(8)
    for n in (10, 11, 12, 13, 31):
        frames = [lb.random_frame(rng, 1, mode="mono", fill=0) for _ in range(n)]      # silent frames always fit: one run
        parsed, want = same_stream(b"".join(frames))
        assert list(parsed[1]) == [(12 * n + 17) // 18] and parsed[0]["pcm_samples"] == 384 * n
