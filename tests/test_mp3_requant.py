"""MP3 quantised front-end: afg_mp3_parse_q records the Huffman values and what the device needs to
requantise them.  Here, on the CPU, a numpy statement of what afg_mp3_requant_hip computes (reference: L3_pow_43
minimp3.d:737-746, the `* sf` of L3_huffman :835-879, stereo processing :885-982, L3_reorder :984-1000) is applied to
those records and must give back, bit for bit, the dequantised spectra of the float front-end -- which the oracle
front-end pins (tests/test_mp3_frontend.py).  The device kernel is held to the same in tests/test_mp3_requant_gpu.py."""
import os

import numpy as np
import pytest

import afgpu
import mp3_bitstream as mb

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "mathjax_invalid_keypress.mp3")


def requant_numpy(q, granules, sdesc):
    """float32 statement of the device kernel: [blocks, 576] spectra."""
    bol, dst, p43 = afgpu.mp3_qtables()
    f32 = np.float32
    out = np.zeros(q.shape, f32)
    for g in granules:
        nch, b0 = int(g["nch"]), int(g["q_off"]) // 576
        x = np.zeros((nch, 576), f32)
        for c in range(nch):
            v = q[b0 + c].astype(np.int64)
            a = np.abs(v)
            one = g["scale"][c][bol[int(g["table"][c]) & 31]]
            p = np.zeros(576, f32)
            small = a < 129
            p[small] = p43[16 + a[small]]
            big = ~small
            if big.any():                                           # L3_pow_43 for x >= 129
                xb = a[big]
                mult = np.where(xb < 1024, 16, 256).astype(f32)
                xb = np.where(xb < 1024, xb << 3, xb)
                sign = (2 * xb) & 64
                frac = ((xb & 63) - sign).astype(f32) / ((xb & ~63) + sign).astype(f32)
                poly = f32(1.0) + frac * (f32(4.0) / f32(3) + frac * (f32(2.0) / f32(9)))
                p[big] = p43[16 + ((xb + sign) >> 6)] * poly * mult
            r = one.astype(f32) * p
            x[c] = np.where(v == 0, f32(0), np.where(v < 0, -r, r))
        if nch == 2 and g["stereo"]:
            mode = np.full(576, 1, np.uint8)
            fl = fr = None
            if g["stereo"] == 2:
                sd = sdesc[int(g["sdesc"])]
                b = bol[int(g["table"][0]) & 31]
                mode, fl, fr = sd["type"][b], sd["fl"][b], sd["fr"][b]
            a0, a1 = x[0].copy(), x[1].copy()
            ms = mode == 1
            x[0][ms], x[1][ms] = (a0 + a1)[ms], (a0 - a1)[ms]
            if fl is not None:
                it = mode == 2
                x[1][it], x[0][it] = (a0 * fr)[it], (a0 * fl)[it]
        for c in range(nch):
            t = int(g["table"][c])
            if t & 0x80:
                out[b0 + c][dst[t & 31]] = x[c]
            else:
                out[b0 + c] = x[c]
    return out


def check_file(data):
    info, runs, q, flags, copies, gr, sd = afgpu.mp3_parse_q(data)
    info2, runs2, coef, flags2, copies2 = afgpu.mp3_parse(data)
    assert info == info2 and np.array_equal(runs, runs2) and np.array_equal(copies, copies2)
    assert q.shape == coef.shape and len(gr) * info["channels"] == len(q)
    assert np.array_equal(flags & 0x00ffffff, flags2 & 0x00ffffff)
    assert ((flags >> 24) >= (flags2 >> 24)).all()              # AFG_MP3_NZ_BANDS: never fewer subbands than the exact bound
    got = requant_numpy(q, gr, sd)
    assert np.array_equal(got.view(np.uint32), coef.view(np.uint32))
    # what the flag promises holds for the quantised path's (possibly wider) bound too
    for blk in range(len(q)):
        nz = int(flags[blk] >> 24) - 1
        assert not got[blk][nz * 18:].view(np.uint32).any()
    return info, gr, sd, q


def test_real_file_requantises_to_the_float_front_end():
    info, gr, sd, q = check_file(open(GOLDEN, "rb").read())
    assert info["channels"] == 2 and (gr["stereo"] == 1).all()          # joint stereo, mid/side on
    assert (gr["table"] & 0x80).any() and int(np.abs(q).max()) > 129     # short blocks and linbits escapes occur


@pytest.mark.parametrize("kw", [
    dict(version="mpeg1", sr=0, mode="ms"), dict(version="mpeg1", sr=1, mode="stereo"), dict(version="mpeg1", sr=2, mode="mono"),
    dict(version="mpeg1", sr=0, mode="intensity"), dict(version="mpeg1", sr=1, mode="ms+intensity"),
    dict(version="mpeg2", sr=0, mode="ms"), dict(version="mpeg2", sr=1, mode="intensity"), dict(version="mpeg2", sr=2, mode="ms+intensity"),
    dict(version="mpeg25", sr=0, mode="stereo"), dict(version="mpeg25", sr=1, mode="intensity"),
])
def test_generated_streams(kw):
    for seed in (1, 2, 3):
        data = mb.make_file(40 + seed, n_frames=10, **kw)[0]
        check_file(data)


def test_mpeg25_8khz_mixed_blocks_are_refused_not_mangled():
    """The one case the device requantiser does not cover: the reference's reorder walks 24 lines outside the channel
    there (minimp3.d:1218-1223).  The quantised parse reports it; the float path still takes the file."""
    refused = 0
    for seed in range(12):
        data = mb.make_file(70 + seed, n_frames=8, version="mpeg25", sr=2, mode="stereo")[0]
        try:
            check_file(data)
        except afgpu.AfgError as e:
            assert "MPEG-2.5" in str(e)
            refused += 1
            afgpu.mp3_parse(data)
    assert refused >= 1


def test_a_mono_frame_with_the_intensity_bit_takes_the_float_path():
    """tests/golden/soak_r05_mono_intensity.mp3 (a damaged generated file, found by tools/soak_damaged.py seed 7): mono frames
    whose header has the intensity-stereo bit set.  The reference runs L3_intensity_stereo on them all the same
    (minimp3.d:100, :1207-1210), which the float front-end restates and the oracle pins; the quantised records cannot say it,
    so afg_mp3_parse_q hands the file to the float path -- it used to record the granules as plain mono and the batch decoded
    1 000 frames of the file to other samples than the stream surface and the oracle."""
    import oraclelib
    data = open(os.path.join(os.path.dirname(__file__), "golden", "soak_r05_mono_intensity.mp3"), "rb").read()
    with pytest.raises(afgpu.AfgError):
        afgpu.mp3_parse_q(data)
    info, runs, coef, flags, copies = afgpu.mp3_parse(data)
    want = oraclelib.mp3_decode_file(data)
    assert info["channels"] == 1 and np.array_equal(coef.view(np.uint32), want["coef"].view(np.uint32))
