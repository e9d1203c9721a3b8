"""Vorbis inverse coupling + floor curves on the device (afg_vorbis_floor_hip) against the oracle
restatement of stb_vorbis2.d:2493-2523 / :2255-2284 / :1534-1563 (oraclelib.vorbis_floor), through the C ABI:
records of real and generated files, and hand-made curves that stress draw_line's integer arithmetic."""
import os

import numpy as np
import pytest

import afgpu
import oraclelib

pytestmark = pytest.mark.gpu
OGG = os.path.join(os.path.dirname(__file__), "golden", "mathjax_invalid_keypress.ogg")


def run_kernel(gpu, packets, curves, points, steps, spec):
    import torch
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(gpu)
    d_spec = torch.from_numpy(spec.copy()).to(gpu)
    d_pk, d_cv = t(packets), t(curves)
    d_pt = t(points if len(points) else np.zeros((1, 2), np.int32))
    d_st = t(steps if len(steps) else np.zeros((1, 2), np.uint8))
    afgpu.vorbis_floor(len(packets), d_pk, d_cv, d_pt, d_st, d_spec)
    torch.cuda.synchronize()
    return d_spec.cpu().numpy()


def concat(parsed):
    """records of several files as one launch (offsets rebased file after file)"""
    pk, cv, pt, st, sp = [], [], [], [], []
    n_cv = n_pt = n_st = n_sp = 0
    for r in parsed:
        k = r["fl_packets"].copy()
        k["spec_off"] += np.uint64(n_sp)
        k["curve_index"] += np.uint32(n_cv)
        k["step_off"] += np.uint32(n_st)
        c = r["fl_curves"].copy()
        c["point_off"] += np.uint32(n_pt)
        pk.append(k); cv.append(c); pt.append(r["fl_points"]); st.append(r["fl_steps"]); sp.append(r["spec"])
        n_cv += len(c); n_pt += len(r["fl_points"]); n_st += len(r["fl_steps"]); n_sp += len(r["spec"])
    return (np.concatenate(pk), np.concatenate(cv), np.concatenate(pt).reshape(-1, 2), np.concatenate(st).reshape(-1, 2), np.concatenate(sp))


def test_records_of_files(gpu):
    import vorbis_bitstream as vb
    files = [open(OGG, "rb").read()]
    for channels, bs in [(1, (256, 2048)), (2, (256, 2048)), (2, (512, 512)), (3, (256, 1024)), (6, (1024, 4096)), (2, (2048, 8192)), (16, (256, 256))]:
        for seed in range(3):
            files.append(vb.make_file(100 * channels + seed, channels=channels, bs=bs, n_packets=20,
                                      residue_types=[(0, 1), (1, 2), (2, 0)][seed]))
    parsed = [afgpu.vorbis_parse_r(f) for f in files]
    full = np.concatenate([afgpu.vorbis_parse(f)["spec"] for f in files])
    pk, cv, pt, st, sp = concat(parsed)
    want = oraclelib.vorbis_floor(pk, cv, pt, st, sp)
    assert np.array_equal(want.view(np.uint32), full.view(np.uint32))             # (the CPU suite's statement, on the joined records)
    got = run_kernel(gpu, pk, cv, pt, st, sp)
    bad = np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0]
    assert len(bad) == 0, (len(bad), bad[:8], got[bad[:8]], want[bad[:8]])
    assert int(pk["n_steps"].sum()) > 0 and (cv["n_points"] == 0).any()


def test_hand_made_curves(gpu):
    import vorbis_floor_cases as cases
    for seed in (5, 6):
        pk, cv, pt, st, spec = cases.hand_made(seed)
        want = oraclelib.vorbis_floor(pk, cv, pt, st, spec)
        cases.same_floats(run_kernel(gpu, pk, cv, pt, st, spec), want)
        assert np.isnan(want).any() or np.isinf(want).any()


def test_batch_and_stream_equal_the_host_floor(gpu, monkeypatch):
    """the outer surface with the floor on the device (default) and in the host parser: the same PCM, bit for bit"""
    import vorbis_bitstream as vb
    from test_stream_gpu import read_all
    files = [open(OGG, "rb").read()] + [vb.make_file(900 + s, channels=1 + s % 3, bs=[(256, 2048), (512, 1024)][s % 2], n_packets=90) for s in range(5)]
    dev = afgpu.batch_decode(files)
    streams = []
    for f in files[:3]:
        s = afgpu.AudioStream()
        s.openFromMemory(f)
        streams.append(read_all(s, s.getNumChannels(), 1000))
        s.cleanUp()
    monkeypatch.setenv("AFG_VORBIS_HOST_FLOOR", "1")
    host = afgpu.batch_decode(files)
    for d, h in zip(dev, host):
        assert d["status"] == 0 == h["status"] and d["frames"] == h["frames"] > 0
        assert np.array_equal(d["pcm"].view(np.uint32), h["pcm"].view(np.uint32))
    for got, h in zip(streams, host):
        assert np.array_equal(got.view(np.uint32).reshape(-1), h["pcm"].view(np.uint32).reshape(-1))
