"""Pins the Vorbis oracle (oracle/vorbis_transform.c) against the float64 IMDCT definition,
the power-complementary window property and time-domain alias cancellation."""
import numpy as np
import pytest

import oraclelib

rng = np.random.default_rng(7)


@pytest.mark.parametrize("n", [256, 512, 1024, 2048, 4096])
def test_inverse_mdct_matches_direct_sum(n):
    X = rng.standard_normal(n // 2).astype(np.float32)
    y = oraclelib.vorbis_inverse_mdct(X, n)
    m = np.arange(n)[:, None]; k = np.arange(n // 2)[None, :]
    want = np.cos(np.pi / (2 * n) * (2 * m + 1 + n / 2) * (2 * k + 1)) @ X.astype(np.float64)
    assert np.abs(y - want).max() < 4e-7 * n ** 0.5 * np.abs(want).max() + 1e-5


def test_tables():
    for n in (256, 2048):
        t = oraclelib.vorbis_tables(n)
        w = t["window"].astype(np.float64)
        assert np.abs(w ** 2 + w[::-1] ** 2 - 1).max() < 1e-6          # power complementary
        k = np.arange(n // 4)
        assert np.abs(t["A"][0::2] - np.cos(4 * k * np.pi / n)).max() < 1e-6
        assert np.abs(t["A"][1::2] + np.sin(4 * k * np.pi / n)).max() < 1e-6
        assert np.abs(t["B"][0::2] - 0.5 * np.cos((2 * k + 1) * np.pi / n / 2)).max() < 1e-6
        ld = n.bit_length() - 1
        rev = [int(format(i, f"0{ld - 3}b")[::-1], 2) << 2 for i in range(n // 8)]
        assert (t["bitrev"] == np.array(rev, np.uint16)).all()


def test_window_bounds_and_layout():
    L, P, N = 1, 2, 4
    pf = np.array([L | P | N, L | P, 0, 0, L | N, L | P | N], np.uint8)
    so, oo, st, ot = oraclelib.vorbis_layout([6], [2], [256], [2048], pf)
    # frames per packet: first none; long->short 1024+448 ... per stb_vorbis2.d:2333-2349
    frames = np.diff(np.append(oo, ot)) // 2
    assert list(frames) == [0, 1472, 128, 128, 576, 1024]
    assert list(np.diff(np.append(so, st)) // 2) == [1024, 1024, 128, 128, 1024, 1024]


def test_tdac_reconstruction_through_finish_frame():
    n, npk = 2048, 6
    sig = rng.standard_normal((npk + 1) * (n // 2))
    w = oraclelib.vorbis_tables(n)["window"].astype(np.float64)
    win = np.concatenate([w, w[::-1]])
    m = np.arange(n)[:, None]; k = np.arange(n // 2)[None, :]
    basis = np.cos(np.pi / (2 * n) * (2 * m + 1 + n / 2) * (2 * k + 1))
    spec = np.stack([(sig[p * n // 2:p * n // 2 + n] * win) @ basis * (2.0 / (n // 2)) for p in range(npk)]).astype(np.float32)
    pf = np.full(npk, 7, np.uint8)
    so, oo, st, ot = oraclelib.vorbis_layout([npk], [1], [256], [n], pf)
    out = oraclelib.vorbis_transform([npk], [1], [256], [n], pf, so, oo, spec.reshape(-1), ot)
    assert np.abs(out - sig[n // 2:n // 2 + out.size]).max() < 2e-4
