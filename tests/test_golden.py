"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py): the oracle must keep
reproducing them on CPU; on the GPU box the HIP path must reproduce them through the C ABI."""
import os

import numpy as np
import pytest

import oraclelib

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(HERE, name))


def same_bits(a, b):
    return a.shape == b.shape and (a.view(np.uint32) == b.view(np.uint32)).all()


def test_oracle_reproduces_mp3_fixture():
    g = load("mp3_transform.npz")
    assert same_bits(oraclelib.mp3_transform(g["granules"], g["channels"], g["coef"], g["flags"]), g["pcm"])


def test_oracle_reproduces_vorbis_fixture():
    g = load("vorbis_transform.npz")
    out = oraclelib.vorbis_transform(g["packets"], g["channels"], g["bs0"], g["bs1"], g["pflags"],
                                     g["spec_off"], g["out_off"], g["spec"], g["out"].size)
    assert same_bits(out, g["out"])


def test_oracle_reproduces_flac_fixture():
    g = load("flac_restore.npz")
    frames = g["frames"].view(oraclelib.FLAC_FRAME_DTYPE)
    sub = g["subframes"].view(oraclelib.FLAC_SUBFRAME_DTYPE)
    oi, of = oraclelib.flac_transform(frames, sub, g["res"], g["out_i32"].size, want_float=True)
    assert (oi == g["out_i32"]).all() and same_bits(of, g["out_f32"])


@pytest.mark.gpu
def test_hip_reproduces_mp3_fixture(gpu):
    import torch
    from afgpu import Mp3Plan
    g = load("mp3_transform.npz")
    plan = Mp3Plan(g["granules"], g["channels"], 4)
    d_pcm = torch.zeros(g["pcm"].size, dtype=torch.float32, device=gpu)
    plan.transform(torch.from_numpy(g["coef"]).to(gpu), torch.from_numpy(g["flags"].view(np.int32)).to(gpu), d_pcm)
    torch.cuda.synchronize()
    got = d_pcm.cpu().numpy()
    rms = float(np.sqrt(np.mean((got.astype(np.float64) - g["pcm"]) ** 2)))
    assert rms <= 1e-5                      # north_star tolerance (float)
    assert same_bits(got, g["pcm"])         # and in fact bit-exact (library built with -ffp-contract=off)


@pytest.mark.gpu
def test_hip_reproduces_vorbis_fixture(gpu):
    import torch
    from afgpu import VorbisPlan
    g = load("vorbis_transform.npz")
    plan = VorbisPlan(g["packets"], g["channels"], g["bs0"], g["bs1"], g["pflags"], 3)
    so, oo = plan.offsets()
    assert (so == g["spec_off"]).all() and (oo == g["out_off"]).all()
    d_out = torch.zeros(g["out"].size, dtype=torch.float32, device=gpu)
    plan.transform(torch.from_numpy(g["spec"]).to(gpu), d_out)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    assert float(np.sqrt(np.mean((got.astype(np.float64) - g["out"]) ** 2))) <= 1e-5
    assert same_bits(got, g["out"])


@pytest.mark.gpu
@pytest.mark.numeric_tolerance
def test_hip_reproduces_vorbis_fixture_within_tolerance(gpu):
    """the default numeric mode: the re-factored inverse MDCT of csrc/vorbis_walk.hip against the frozen oracle output"""
    import torch
    from afgpu import VorbisPlan
    g = load("vorbis_transform.npz")
    plan = VorbisPlan(g["packets"], g["channels"], g["bs0"], g["bs1"], g["pflags"], 3)
    d_out = torch.full((g["out"].size,), float("nan"), dtype=torch.float32, device=gpu)
    plan.transform(torch.from_numpy(g["spec"]).to(gpu), d_out)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    assert not np.isnan(got).any()
    assert float(np.sqrt(np.mean((got.astype(np.float64) - g["out"]) ** 2))) <= 1e-5


@pytest.mark.gpu
def test_hip_reproduces_flac_fixture(gpu):
    import torch
    import afgpu
    g = load("flac_restore.npz")
    n = g["frames"].size // 32
    d_i = torch.zeros(g["out_i32"].size, dtype=torch.int32, device=gpu)
    d_f = torch.zeros(g["out_i32"].size, dtype=torch.float32, device=gpu)
    afgpu.flac_transform(n, torch.from_numpy(g["frames"]).to(gpu), torch.from_numpy(g["subframes"]).to(gpu),
                         torch.from_numpy(g["res"]).to(gpu), d_i, d_f)
    torch.cuda.synchronize()
    assert (d_i.cpu().numpy() == g["out_i32"]).all()                   # bit-exact int32
    assert same_bits(d_f.cpu().numpy(), g["out_f32"])


def test_oracle_reproduces_opus_file_fixture():
    g = load("opus_file.npz")
    rec = oraclelib.opus_decode_file(g["data"].tobytes())
    assert rec["gain_i"] == int(g["gain_i"]) and rec["declared_frames"] == int(g["declared"])
    assert same_bits(rec["coeffs"], g["coeffs"]) and rec["frames"].tobytes() == g["frames"].tobytes()
    assert same_bits(oraclelib.opus_file_pcm(rec), g["pcm"])


def test_product_front_end_reproduces_opus_file_fixture():
    import afgpu
    g = load("opus_file.npz")
    got = afgpu.opus_parse(g["data"].tobytes())
    assert same_bits(got["coeffs"], g["coeffs"]) and got["frames"].tobytes() == g["frames"].tobytes()


def test_oracle_reproduces_layer2_file_fixture():
    g = load("mp2_file.npz")
    want = oraclelib.mp3_decode_file(g["data"].tobytes())
    assert want["layer"] == 2 and (want["channels"], want["hz"], want["declared_samples"]) == (int(g["channels"]), int(g["hz"]), int(g["declared"]))
    assert same_bits(want["pcm"], g["pcm"]) and len(g["pcm"]) > 0


@pytest.mark.gpu
def test_hip_reproduces_opus_and_layer2_file_fixtures(gpu):
    import afgpu
    o, m = load("opus_file.npz"), load("mp2_file.npz")
    res = afgpu.batch_decode([o["data"].tobytes(), m["data"].tobytes()])
    assert [r["status"] for r in res] == [0, 0]
    assert same_bits(res[0]["pcm"], o["pcm"])
    assert same_bits(res[1]["pcm"].reshape(-1), m["pcm"])


def floor_fixture():
    import afgpu
    g = load("vorbis_floor.npz")
    return (g["packets"].view(afgpu.VORBIS_FLOOR_PACKET_DTYPE), g["curves"].view(afgpu.VORBIS_FLOOR_CURVE_DTYPE), g["points"], g["steps"],
            g["residue"], g["spec"])


def test_oracle_reproduces_vorbis_floor_fixture():
    """records of the product's parser + oracle restatement of coupling / do_floor == the oracle front-end's spectra (real file)"""
    pk, cv, pt, st, residue, spec = floor_fixture()
    assert len(pk) == 7 and (cv["n_points"] == 0).any() and not same_bits(residue, spec)
    assert same_bits(oraclelib.vorbis_floor(pk, cv, pt, st, residue), spec)


@pytest.mark.gpu
def test_hip_reproduces_vorbis_floor_fixture(gpu):
    import torch
    import afgpu
    pk, cv, pt, st, residue, spec = floor_fixture()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(gpu)
    d_spec = torch.from_numpy(residue.copy()).to(gpu)
    afgpu.vorbis_floor(len(pk), t(pk), t(cv), t(pt), t(st), d_spec)
    torch.cuda.synchronize()
    assert same_bits(d_spec.cpu().numpy(), spec)


def rows16_fixture():
    g, p = load("flac_restore.npz"), load("flac_rows16.npz")
    return (p["frames"].view(oraclelib.FLAC_FRAME_DTYPE), g["subframes"].view(oraclelib.FLAC_SUBFRAME_DTYPE), p["res"], g["out_i32"], g["out_f32"])


def test_oracle_reproduces_flac_fixture_from_int16_rows():
    frames, sub, res, want_i, want_f = rows16_fixture()
    assert frames["res16"].any() and not frames["res16"].all()
    oi, of = oraclelib.flac_transform(frames, sub, res, want_i.size, want_float=True)
    assert (oi == want_i).all() and same_bits(of, want_f)


@pytest.mark.gpu
def test_hip_reproduces_flac_fixture_from_int16_rows(gpu):
    import torch
    import afgpu
    frames, sub, res, want_i, want_f = rows16_fixture()
    d_i = torch.zeros(want_i.size, dtype=torch.int32, device=gpu)
    d_f = torch.zeros(want_i.size, dtype=torch.float32, device=gpu)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(gpu)
    d_res = torch.from_numpy(res.copy()).to(gpu)
    for variants in (None, afgpu.flac_variants(frames, sub)):
        d_i.zero_(); d_f.zero_()
        afgpu.flac_transform(len(frames), dev(frames), dev(sub), d_res, d_i, d_f, None, variants=variants)
        torch.cuda.synchronize()
        assert (d_i.cpu().numpy() == want_i).all() and same_bits(d_f.cpu().numpy(), want_f)
