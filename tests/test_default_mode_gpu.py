"""The outer surface in the DEFAULT numeric mode (AFG_NUMERIC_TOLERANCE): what a user of the library gets without setting
anything.  The bit-exact suites (test_stream_gpu.py, test_multidevice_gpu.py, ...) pin AFG_NUMERIC=exact; here the same paths
-- AudioStream pulls, seeks, afg_batch_decode over mixed batches, several device entries, a C5 wave -- run as shipped:
FLAC / QOA stay bit-identical to the oracle, MP3 (csrc/mp3_tolerance.hip), Ogg Vorbis (csrc/vorbis_walk.hip) and Ogg Opus
(csrc/celt_walk.hip) within north_star's 1e-5 RMS, and everything that must not depend on HOW a file was decoded (chunked pulls against the
batch decode, the device list, the sharding) is still compared bit for bit."""
import numpy as np
import pytest

import afgpu
import flac_bitstream as fb
import oraclelib
from test_flac_frontend import make_pcm
from test_stream_gpu import MP3_FIXTURE, OGG_FIXTURE, long_files, qoa_file, read_all

pytestmark = [pytest.mark.gpu, pytest.mark.numeric_tolerance]

TOL = 1e-5
ONE_STEP = 1 / 32767 + 1.2e-7          # neighbouring int16 values as float32 / 32767 (6e-8 apart near full scale)


def rms(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


def test_the_default_mode_is_tolerance(gpu):
    assert afgpu.get_numeric_mode() == afgpu.NUMERIC_TOLERANCE


def test_ogg_stream_default_mode(gpu):
    data = open(OGG_FIXTURE, "rb").read()
    want = oraclelib.vorbis_file_pcm(oraclelib.vorbis_decode_file(data))
    s = afgpu.AudioStream()
    s.openFromMemory(data)
    assert not s.isError(), s.errorMessage()
    assert s.getFormat() == afgpu.FORMAT_OGG and s.getNumChannels() == 2 and s.getLengthInFrames() == 22050
    got = read_all(s, 2, 1000)
    assert got.shape == want.shape and not np.isnan(got).any()
    assert rms(got, want) <= TOL
    assert not np.array_equal(got.view(np.uint32), want.view(np.uint32)), "the exact kernel ran in the default mode"


def test_all_formats_in_one_batch_default_mode(gpu):
    ogg = open(OGG_FIXTURE, "rb").read()
    mp3 = open(MP3_FIXTURE, "rb").read()
    flac, _ = fb.encode_file(make_pcm(5000, 2, 16, 8), 16, 1024)
    qoa, qoa_want = qoa_file(7000, 2, 44100, 9)
    import opus_bitstream
    opus, _ = opus_bitstream.random_celt_file(np.random.default_rng(8), 2, 30, preskip=312, comments=(b"R128_TRACK_GAIN=-19000",))
    files = [ogg, mp3, flac, qoa, opus, b"junk" * 100, ogg]
    out = afgpu.batch_decode(files, n_threads=4)
    assert [o["format"] for o in out[:5]] == [afgpu.FORMAT_OGG, afgpu.FORMAT_MP3, afgpu.FORMAT_FLAC, afgpu.FORMAT_QOA, afgpu.FORMAT_OPUS]
    assert out[5]["status"] != 0
    want_ogg = oraclelib.vorbis_file_pcm(oraclelib.vorbis_decode_file(ogg))
    for k in (0, 6):
        assert out[k]["frames"] == len(want_ogg) and rms(out[k]["pcm"], want_ogg) <= TOL
    assert np.array_equal(out[0]["pcm"].view(np.uint32), out[6]["pcm"].view(np.uint32))          # the same file twice: the same bits
    want_mp3 = oraclelib.mp3_decode_file(mp3)["pcm"]
    assert out[1]["pcm"].size == want_mp3.size and rms(out[1]["pcm"].reshape(-1), want_mp3) <= TOL
    assert np.array_equal(out[3]["pcm"].view(np.uint32), qoa_want.view(np.uint32))
    info, frames, subs, res = afgpu.flac_parse(flac)
    want_flac = oraclelib.flac_transform(frames, subs, res, info["out_samples"], want_float=True)[1]
    assert np.array_equal(out[2]["pcm"].reshape(-1).view(np.uint32), want_flac.view(np.uint32))
    want_opus = oraclelib.opus_file_pcm(oraclelib.opus_decode_file(opus))
    step = np.abs(out[4]["pcm"].astype(np.float64) - want_opus)
    assert out[4]["frames"] == len(want_opus) and step.max() <= ONE_STEP and (step > 0).mean() < 0.01


@pytest.mark.parametrize("kind", ["ogg", "mp3", "flac"])
def test_chunked_reads_equal_the_batch_decode_default_mode(gpu, kind):
    """A stream decodes 64 packets at a time as the caller pulls, the batch path whole files in 16-packet segments: in the
    default mode too the delivered samples do not depend on that (the walk's result is independent of the segmentation)."""
    data = long_files()[kind]
    want = afgpu.batch_decode([data])[0]
    assert want["status"] == 0 and want["frames"] > 20000
    ch = want["channels"]
    for chunk in (1024, 777):
        s = afgpu.AudioStream()
        s.openFromMemory(data)
        assert not s.isError(), s.errorMessage()
        got = read_all(s, ch, chunk)
        assert got.shape == (want["frames"], ch)
        assert np.array_equal(got.view(np.uint32), want["pcm"].view(np.uint32))
        s.cleanUp()
    if kind == "ogg":
        ref = oraclelib.vorbis_file_pcm(oraclelib.vorbis_decode_file(data))
        assert ref.shape == want["pcm"].shape and rms(want["pcm"], ref) <= TOL          # absolute: the generated files sit inside full scale (round 5)


def test_ogg_seeks_default_mode(gpu):
    data = long_files()["ogg"]
    want = afgpu.batch_decode([data])[0]
    ch, total = want["channels"], want["frames"]
    s = afgpu.AudioStream()
    s.openFromMemory(data)
    for target in [total // 2, 10, total - 7, total // 3, 0, total // 3 + 1]:
        assert s.seekPosition(int(target)) and s.tellPosition() == target
        buf = np.zeros(500 * ch, np.float32)
        got = s.readSamplesFloat(buf)
        assert got == min(500, total - target)
        assert np.array_equal(buf[:got * ch].view(np.uint32), want["pcm"][target:target + got].reshape(-1).view(np.uint32))
    s.cleanUp()


def test_device_list_does_not_change_the_samples_default_mode(gpu):
    import vorbis_bitstream as vb
    files = [vb.make_file(300 + k, n_packets=40 + 7 * k) for k in range(5)] + [open(MP3_FIXTURE, "rb").read()]
    one = afgpu.batch_decode(files)
    assert all(o["status"] == 0 for o in one)
    for devices in ([0, 0], [0, 0, 0], "all"):
        many = afgpu.batch_decode(files, devices=devices)
        for a, b in zip(one, many):
            assert (a["status"], a["frames"], a["channels"]) == (b["status"], b["frames"], b["channels"])
            assert np.array_equal(a["pcm"].view(np.uint32), b["pcm"].view(np.uint32))


def test_c5_wave_default_mode(gpu):
    """a wave of the mixed corpus with all four codecs resident together: FLAC bit-exact, MP3 / Vorbis / CELT within tolerance,
    and the file results independent of which wave (shard) a file lands in"""
    import torch
    from afgpu import corpus
    from test_multidevice_gpu import oracle_file_outputs, small_manifest
    man = small_manifest(48)
    ids = np.arange(48)

    def decode(sel):
        wl = corpus.build_c5_wave(man, sel, gpu, host=True)
        wl.step(torch.cuda.current_stream())
        torch.cuda.synchronize()
        got = {}
        for part in wl.parts:
            for fid, arr in zip(part.file_ids, part.file_outputs()):
                got[int(fid)] = (part.name, arr.copy())
        return got, wl
    whole, wl = decode(ids)
    for part in wl.parts:
        for fid, want in zip(part.file_ids, oracle_file_outputs(part)):
            name, got = whole[int(fid)]
            if name == "flac":
                assert np.array_equal(got.view(np.uint32), np.asarray(want).view(np.uint32)), (name, fid)
            else:
                assert not np.isnan(got).any() and rms(got, want) <= TOL, (name, fid)
    for shard in (ids[0::2], ids[1::2]):
        part_got, _ = decode(shard)
        for fid, (name, arr) in part_got.items():
            assert np.array_equal(arr.view(np.uint32), whole[fid][1].view(np.uint32)), (name, fid)


def test_generated_vorbis_stream_shapes_default_mode(gpu):
    """Generated Ogg files of every stream shape the tolerance-mode walk has a kernel for (mono, 1024- and 4096-sample long
    blocks, three .. six channels) and of two it has not, from bytes to PCM in one batch: within 1e-5 RMS (absolute: the
    generator is calibrated to rms 0.05) of the oracle's decode, and not its bits where the walk ran."""
    import vorbis_bitstream as vb
    shapes = [(1, (256, 2048)), (2, (256, 1024)), (1, (256, 1024)), (2, (512, 4096)), (1, (256, 4096)), (3, (256, 1024)),
              (6, (256, 2048)), (4, (512, 4096)), (5, (512, 2048)), (2, (1024, 2048)), (2, (2048, 8192))]
    files, wants = [], []
    for k, (ch, bs) in enumerate(shapes):
        d = vb.make_file(1700 + k, channels=ch, bs=bs, n_packets=30, residue_types=[(0, 1), (1, 2), (2, 0)][k % 3])
        rec = oraclelib.vorbis_decode_file(d)
        assert rec is not None and len(rec["pflags"]) >= 20
        files.append(d)
        wants.append(oraclelib.vorbis_file_pcm(rec))
    out = afgpu.batch_decode(files, n_threads=2)
    walked = 0
    for (ch, bs), item, want in zip(shapes, out, wants):
        assert item["status"] == 0 and item["frames"] == len(want) and item["channels"] == ch
        assert rms(item["pcm"], want) <= 1e-5, (ch, bs)
        same = np.array_equal(item["pcm"].view(np.uint32), want.view(np.uint32))
        if bs[0] <= 512 and bs[1] in (1024, 2048, 4096):
            walked += not same
        else:
            assert same, (ch, bs)                # the bit-exact kernels
    assert walked >= 7


def test_grouped_batch_pipeline_does_not_change_the_samples(gpu):
    """A large batch call runs as a pipeline of groups of files (afg.h: dev option "batch_groups"; by default from 512 files up):
    every sample is the one the ungrouped call delivers -- all formats, a broken file and an empty one in the middle, a group
    count that does not divide the batch."""
    import opus_bitstream
    import vorbis_bitstream as vb
    rng = np.random.default_rng(61)
    kinds = []
    for k in range(6):
        flac, _ = fb.encode_file(make_pcm(3000 + 500 * k, 2 if k % 2 else 1, 16, 30 + k), 16, 576 if k % 2 else 1024)
        qoa, _ = qoa_file(5000 + 300 * k, 1 + k % 2, 44100, 40 + k)
        opus, _ = opus_bitstream.random_celt_file(np.random.default_rng(50 + k), 1 + k % 2, 8 + k, preskip=312)
        kinds += [flac, qoa, opus, vb.make_file(400 + k, n_packets=12 + 3 * k)]
    kinds += [open(MP3_FIXTURE, "rb").read(), open(OGG_FIXTURE, "rb").read(), b"junk" * 64, b""]
    files = [bytes(kinds[int(i)]) for i in rng.integers(0, len(kinds), 520)]
    import os
    old = os.environ.get("AFG_BATCH_GROUPS")
    try:
        os.environ["AFG_BATCH_GROUPS"] = "1"
        one = afgpu.batch_decode(files, n_threads=6)
        for groups in ("2", "5", "7"):
            os.environ["AFG_BATCH_GROUPS"] = groups
            many = afgpu.batch_decode(files, n_threads=6)
            for a, b in zip(one, many):
                assert (a["status"], a["format"], a["frames"], a["channels"], a["samplerate"]) == (b["status"], b["format"], b["frames"], b["channels"], b["samplerate"])
                if a["pcm"] is not None:
                    assert np.array_equal(a["pcm"].view(np.uint32), b["pcm"].view(np.uint32))
        os.environ.pop("AFG_BATCH_GROUPS")
        auto = afgpu.batch_decode(files, n_threads=6)            # the library's own choice (a mixed batch: not grouped)
        for a, b in zip(one, auto):
            assert a["status"] == b["status"] and a["frames"] == b["frames"]
            if a["pcm"] is not None:
                assert np.array_equal(a["pcm"].view(np.uint32), b["pcm"].view(np.uint32))
    finally:
        if old is None:
            os.environ.pop("AFG_BATCH_GROUPS", None)
        else:
            os.environ["AFG_BATCH_GROUPS"] = old
    assert sum(o["status"] == 0 for o in one) >= 400 and any(o["status"] != 0 for o in one)
