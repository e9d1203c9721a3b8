"""MPEG Layer I / II end to end on the device: file bytes -> afg_batch_decode / afg_open_from_memory -> PCM, against the
oracle's reference-shaped drive (12-slot synthesis granules, minimp3.d:1557-1578).  The device runs the same synthesis
over the slots packed into 18-slot blocks flagged AFG_MP3_SUBBAND."""
import numpy as np
import pytest

import afgpu
import mp3_l12_bitstream as lb
import oraclelib
from test_stream_gpu import read_all

pytestmark = pytest.mark.gpu


def files():
    rng = np.random.default_rng(31)
    out = []
    for layer, version, sr, mode in [(2, "mpeg1", 0, "stereo"), (2, "mpeg1", 1, "joint"), (2, "mpeg2", 2, "mono"), (1, "mpeg1", 0, "stereo"),
                                     (1, "mpeg2", 1, "mono"), (2, "mpeg1", 2, "dual"), (1, "mpeg1", 2, "joint")]:
        out.append(lb.random_file(rng, layer, 70, version, sr=sr, mode=mode, vary_bitrate=True))
    out.append(b"".join(lb.layer1_frame(rng, "mpeg1", 12, 0, "stereo", 0, quiet=False)[0] for _ in range(40)))
    return out


def test_batch_decode_matches_the_reference_shaped_drive(gpu):
    import mp3_bitstream as mb
    data = files()
    mixed = data + [mb.make_file(77, n_frames=20)[0]]                     # Layer III next to them: quantised upload falls back as a whole
    res = afgpu.batch_decode(mixed)
    for r, d in zip(res, mixed):
        want = oraclelib.mp3_decode_file(d)
        assert r["status"] == 0 and r["format"] == afgpu.FORMAT_MP3
        assert r["channels"] == want["channels"] and r["samplerate"] == want["hz"]
        pcm = want["pcm"].reshape(-1, want["channels"])
        assert r["frames"] == len(pcm) > 0
        assert np.array_equal(r["pcm"].view(np.uint32), pcm.view(np.uint32))
    loud = res[len(data) - 1]["pcm"]
    assert np.isfinite(loud).all() and np.abs(loud).max() > 0.01


@pytest.mark.parametrize("which", [0, 3, 4, 7])
def test_stream_reads_cross_chunk_boundaries(gpu, which):
    """a decode chunk is 64 frames; Layer I frames are two thirds of a block, so a chunk runs on until its slots fill whole
    blocks and the device state carries over"""
    d = files()[which]
    want = oraclelib.mp3_decode_file(d)
    pcm = want["pcm"].reshape(-1, want["channels"])
    for chunk in (1024, 555, 100000):
        s = afgpu.AudioStream()
        s.openFromMemory(d)
        assert not s.isError(), s.errorMessage()
        assert s.getFormat() == afgpu.FORMAT_MP3 and s.getNumChannels() == want["channels"]
        assert s.getLengthInFrames() == want["declared_samples"] // want["channels"]
        got = read_all(s, want["channels"], chunk)
        assert not s.isError()
        assert got.shape == pcm.shape
        assert np.array_equal(got.view(np.uint32), pcm.view(np.uint32))
        s.cleanUp()


def test_subband_flag_skips_the_layer3_tail_on_the_device(gpu):
    """transform level: blocks flagged AFG_MP3_SUBBAND against the oracle (which skips antialias / IMDCT / sign change for
    them), next to ordinary Layer III blocks of other streams in the same launch"""
    import torch
    from afgpu import synthetic
    granules, channels = [30, 13, 50], [2, 1, 2]
    coef, flags = synthetic.mp3_batch(9, granules, channels, p_event=0.2)
    flags = flags.copy()
    blk = np.cumsum([0] + [g * c for g, c in zip(granules, channels)])
    flags[blk[1]:blk[2]] |= np.uint32(0x80000000)                          # the mono stream
    flags[blk[2]:blk[3]] = np.uint32(0x80000000)                           # the last stereo stream, no NZ declaration
    want = oraclelib.mp3_transform(granules, channels, coef, flags)
    plan = afgpu.Mp3Plan(granules, channels, 8)
    d_pcm = torch.zeros(coef.size, dtype=torch.float32, device=gpu)
    plan.transform(torch.from_numpy(coef).to(gpu), torch.from_numpy(flags.view(np.int32)).to(gpu), d_pcm)
    torch.cuda.synchronize()
    assert np.array_equal(d_pcm.cpu().numpy().view(np.uint32), want.view(np.uint32))
