"""afg_mp3_requant_hip on the device: int16 Huffman values + band scales + stereo plan -> the dequantised, stereo-processed,
reordered spectra of the float front-end, bit for bit (minimp3.d:722-746, :835-879, :885-1000); then through the
transform stage to the PCM the oracle decodes from the same file."""
import os

import numpy as np
import pytest

import afgpu
import mp3_bitstream as mb
import oraclelib

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "mathjax_invalid_keypress.mp3")


def device_requant(gpu, q, gr, sd):
    import torch
    d_q = torch.from_numpy(q.reshape(-1).copy()).to(gpu)
    d_gr = torch.from_numpy(gr.view(np.uint8).copy()).to(gpu)
    d_sd = torch.from_numpy(sd.view(np.uint8).copy()).to(gpu) if len(sd) else None
    d_coef = torch.full((q.size,), float("nan"), dtype=torch.float32, device=gpu)
    afgpu.mp3_requant(len(gr), d_gr, d_q, d_sd, d_coef)
    torch.cuda.synchronize()
    return d_coef


def files():
    out = [open(GOLDEN, "rb").read()]
    for kw in (dict(version="mpeg1", sr=0, mode="ms"), dict(version="mpeg1", sr=1, mode="ms+intensity"), dict(version="mpeg1", sr=2, mode="mono"),
               dict(version="mpeg2", sr=1, mode="intensity"), dict(version="mpeg25", sr=0, mode="stereo"), dict(version="mpeg2", sr=2, mode="ms+intensity")):
        for seed in (1, 2, 3):
            out.append(mb.make_file(90 + seed, n_frames=12, **kw)[0])
    return out


def test_requantised_spectra_are_the_float_front_ends(gpu):
    for data in files():
        info, runs, q, flags, copies, gr, sd = afgpu.mp3_parse_q(data)
        _, _, coef, _, _ = afgpu.mp3_parse(data)
        got = device_requant(gpu, q, gr, sd).cpu().numpy().reshape(-1, 576)
        assert not np.isnan(got).any()
        assert np.array_equal(got.view(np.uint32), coef.view(np.uint32))


def test_quantised_path_end_to_end_equals_the_oracle_decode(gpu):
    import torch
    for data in files()[:8]:
        want = oraclelib.mp3_decode_file(data)
        info, runs, q, flags, copies, gr, sd = afgpu.mp3_parse_q(data)
        d_coef = device_requant(gpu, q, gr, sd)
        plan = afgpu.Mp3Plan(runs, np.full(len(runs), info["channels"], np.uint8))
        d_pcm = torch.zeros_like(d_coef)
        plan.transform(d_coef, torch.from_numpy(flags.view(np.int32)).to(gpu), d_pcm)
        torch.cuda.synchronize()
        plane = d_pcm.cpu().numpy()
        pcm = np.concatenate([plane[int(s):int(s) + int(c)] for s, c in copies]) if len(copies) else np.zeros(0, np.float32)
        assert np.array_equal(pcm.view(np.uint32), want["pcm"].view(np.uint32))


def test_batch_path_uploads_quantised_values_and_matches_the_float_upload(gpu, monkeypatch):
    """afg_batch_decode ships int16 values by default; AFG_MP3_FLOAT_UPLOAD=1 ships round 1's float spectra.  Same PCM, and
    the oracle's."""
    blobs = files()
    monkeypatch.delenv("AFG_MP3_FLOAT_UPLOAD", raising=False)
    a = afgpu.batch_decode(blobs, n_threads=4)
    monkeypatch.setenv("AFG_MP3_FLOAT_UPLOAD", "1")
    b = afgpu.batch_decode(blobs, n_threads=4)
    for blob, x, y in zip(blobs, a, b):
        want = oraclelib.mp3_decode_file(blob)
        assert x["status"] == 0 and y["status"] == 0 and x["frames"] == y["frames"] == len(want["pcm"]) // x["channels"]
        if x["frames"]:
            assert np.array_equal(x["pcm"].reshape(-1).view(np.uint32), want["pcm"].view(np.uint32))
            assert np.array_equal(y["pcm"].reshape(-1).view(np.uint32), want["pcm"].view(np.uint32))


def test_a_file_outside_the_requantisers_coverage_sends_the_batch_down_the_float_path(gpu, monkeypatch):
    monkeypatch.delenv("AFG_MP3_FLOAT_UPLOAD", raising=False)
    odd = None
    for seed in range(40):
        data = mb.make_file(70 + seed, n_frames=8, version="mpeg25", sr=2, mode="stereo")[0]
        try:
            afgpu.mp3_parse_q(data)
        except afgpu.AfgError:
            odd = data
            break
    assert odd is not None
    blobs = files()[:4] + [odd] + files()[4:7]
    got = afgpu.batch_decode(blobs, n_threads=4)
    for blob, x in zip(blobs, got):
        want = oraclelib.mp3_decode_file(blob)
        assert x["status"] == 0 and x["frames"] * x["channels"] == len(want["pcm"])
        if x["frames"]:
            assert np.array_equal(x["pcm"].reshape(-1).view(np.uint32), want["pcm"].view(np.uint32))


def test_a_mono_frame_with_the_intensity_bit_in_a_batch(gpu, monkeypatch):
    """tests/golden/soak_r05_mono_intensity.mp3 (tests/test_mp3_requant.py) among ordinary files: the batch takes the float
    path and every file decodes to the oracle's samples."""
    import os
    monkeypatch.delenv("AFG_MP3_FLOAT_UPLOAD", raising=False)
    odd = open(os.path.join(os.path.dirname(__file__), "golden", "soak_r05_mono_intensity.mp3"), "rb").read()
    blobs = files()[:3] + [odd] + files()[3:6]
    got = afgpu.batch_decode(blobs, n_threads=4)
    for blob, x in zip(blobs, got):
        want = oraclelib.mp3_decode_file(blob)
        assert x["status"] == 0 and x["frames"] * x["channels"] == len(want["pcm"])
        if x["frames"]:
            assert np.array_equal(x["pcm"].reshape(-1).view(np.uint32), want["pcm"].view(np.uint32))
    alone = afgpu.batch_decode([odd])[0]
    assert np.array_equal(alone["pcm"].view(np.uint32), got[3]["pcm"].view(np.uint32))
