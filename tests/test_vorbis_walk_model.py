"""The factorisation vorbis_walk.hip runs (csrc/vorbis_walk.hip, AFG_NUMERIC_TOLERANCE), restated on [64 lanes][R] arrays in
tests/vorbis_walk_model.py and checked on the CPU: every FFT size against a library FFT, the whole inverse MDCT against the
defining sum and against the oracle's restatement of stb_vorbis' inverse_mdct (stb_vorbis2.d:1941-2242), the
overlap-add written on the DCT-IV against window * y, and every LDS access of the transposes for bank conflicts and for
staying inside the area the kernel reserves."""
import numpy as np
import pytest

import oraclelib
import vorbis_walk_model as M

AREA = {4: 384, 8: 576, 16: 1152}          # Geo<R>::kChanF2


@pytest.mark.parametrize("R", [4, 8, 16])
def test_fft_passes_match_a_library_fft(R):
    rng = np.random.default_rng(R)
    t = rng.standard_normal((64, R)) + 1j * rng.standard_normal((64, R))
    Z, lds = M.FFT[R](t)
    q = M.group_of(M.LANES)[:, None] + 64 * np.arange(R)[None, :]
    full = np.zeros(64 * R, complex)
    full[q] = t
    assert np.abs(Z - np.fft.fft(full)[q]).max() < 1e-11
    assert lds.worst == 1, "an LDS access of a transpose has a bank conflict"
    assert lds.high < AREA[R]


@pytest.mark.parametrize("R", [4, 8, 16])
def test_inverse_mdct_matches_the_definition_and_the_oracle(R):
    n = 256 * R
    rng = np.random.default_rng(10 + R)
    X = rng.standard_normal(n // 2)
    y, _, _ = M.imdct_model(X, R)
    assert np.abs(y - M.imdct_reference(X, n)).max() < 1e-9
    # the oracle: one long packet of a mono stream, windowed by vorbis_finish_frame -- compare where the window is known
    flags = np.full(2, 7, np.uint8)          # AFG_VORBIS_LONG | _PREV | _NEXT (afg.h)
    X2 = rng.standard_normal(n // 2)
    spec = np.concatenate([X, X2]).astype(np.float32)
    so = np.array([0, n // 2], np.uint64)
    oo = np.array([0, 0], np.uint64)
    got = oraclelib.vorbis_transform([2], [1], [256], [n], flags, so, oo, spec, n // 2)
    y0, c0, _ = M.imdct_model(spec[:n // 2].astype(np.float64), R)
    y1, c1, _ = M.imdct_model(spec[n // 2:].astype(np.float64), R)
    win = M.vorbis_window(n)
    jj = np.arange(n // 2)
    want = y0[n // 2 + jj] * win[n // 2 - 1 - jj] + y1[jj] * win[jj]
    assert np.abs(got - want).max() < 2e-5 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("R", [4, 8, 16])
def test_overlap_add_on_the_dct_iv(R):
    n = 256 * R
    rng = np.random.default_rng(20 + R)
    X0, X1 = rng.standard_normal(n // 2), rng.standard_normal(n // 2)
    y0, c0, _ = M.imdct_model(X0, R)
    y1, c1, _ = M.imdct_model(X1, R)
    win = M.vorbis_window(n)
    _, cb = M.tdac_frames(c0, np.zeros((64, R)), win, R)
    out, _ = M.tdac_frames(c1, cb, win, R)
    jj = np.arange(n // 2)
    want = y0[n // 2 + jj] * win[n // 2 - 1 - jj] + y1[jj] * win[jj]
    assert np.abs(out - want).max() < 1e-12
