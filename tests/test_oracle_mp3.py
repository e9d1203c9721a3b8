"""Pins the MP3 oracle (oracle/mp3_transform.c) against float64 textbook definitions.

The reference ships no golden vectors, so the restatement is pinned by
the ISO 11172-3 definitions instead: direct IMDCT / DCT-II sums, perfect reconstruction through
the window-switching hybrid filterbank, and the modulation structure of the polyphase bank."""
import numpy as np

import oraclelib

L = oraclelib.lib()
rng = np.random.default_rng(2026)


def iso_window(bt):
    i = np.arange(36)
    w = np.zeros(36)
    if bt == 0:
        w = np.sin(np.pi / 36 * (i + 0.5))
    elif bt == 1:
        w[:18] = np.sin(np.pi / 36 * (i[:18] + 0.5)); w[18:24] = 1; w[24:30] = np.sin(np.pi / 12 * (i[24:30] - 18 + 0.5))
    elif bt == 3:
        w[6:12] = np.sin(np.pi / 12 * (i[6:12] - 6 + 0.5)); w[12:18] = 1; w[18:] = np.sin(np.pi / 36 * (i[18:] + 0.5))
    return w


def forward_mdct(z, bt):
    """ISO forward MDCT of 36 subband samples (previous 18 + current 18), minimp3 line order."""
    if bt != 2:
        i = np.arange(36)[:, None]; k = np.arange(18)[None, :]
        return (z * iso_window(bt)) @ np.cos(np.pi / 72 * (2 * i + 1 + 18) * (2 * k + 1)) / 9.0
    X = np.zeros(18)
    i = np.arange(12)[:, None]; k = np.arange(6)[None, :]
    ws = np.sin(np.pi / 12 * (np.arange(12) + 0.5))
    for w in range(3):
        X[3 * np.arange(6) + w] = (z[6 + 6 * w:18 + 6 * w] * ws) @ np.cos(np.pi / 24 * (2 * i + 1 + 6) * (2 * k + 1)) / 3.0
    return X


def test_imdct36_matches_direct_sum():
    X0 = rng.standard_normal(576).astype(np.float32)
    X1 = rng.standard_normal(576).astype(np.float32)
    ov = np.zeros(288, np.float32)
    g = X0.copy(); L.afgo_mp3_imdct_gr(g, ov, 0, 0)
    g = X1.copy(); L.afgo_mp3_imdct_gr(g, ov, 0, 0)
    m = np.arange(36)[:, None]; k = np.arange(18)[None, :]
    M = np.cos(np.pi / 72 * (2 * m + 19) * (2 * k + 1))
    w = np.sin(np.pi / 36 * (np.arange(36) + 0.5))
    cur = (M @ X1.reshape(32, 18).T.astype(np.float64)).T
    prv = (M @ X0.reshape(32, 18).T.astype(np.float64)).T
    want = w[:18] * cur[:, :18] + w[18:] * prv[:, 18:]
    assert np.abs(g.reshape(32, 18) - want).max() < 2e-5


def test_window_switching_reconstructs_signal():
    """long -> start -> short -> stop -> long through the oracle reproduces the subband signal
    (time-domain alias cancellation with the ISO window shapes), scale exactly +1."""
    types = [0, 0, 1, 2, 2, 3, 0, 1, 2, 3, 0, 0]
    sig = rng.standard_normal((32, 18 * (len(types) + 1)))
    ov = np.zeros(288, np.float32)
    out = np.zeros((32, 18 * len(types)))
    for g, bt in enumerate(types):
        X = np.stack([forward_mdct(sig[b, 18 * g:18 * g + 36], bt) for b in range(32)])
        gr = X.astype(np.float32).reshape(-1).copy()
        L.afgo_mp3_imdct_gr(gr, ov, bt, 0)
        out[:, 18 * g:18 * g + 18] = gr.reshape(32, 18)
    assert np.abs(out[:, 18:] - sig[:, 18:18 * len(types)]).max() < 5e-6


def test_mixed_block_uses_long_transform_below_n_long_bands():
    types_a = np.random.default_rng(5).standard_normal(576).astype(np.float32)
    ov1 = np.zeros(288, np.float32); ov2 = np.zeros(288, np.float32)
    a = types_a.copy(); L.afgo_mp3_imdct_gr(a, ov1, 2, 2)      # mixed: bands 0,1 long
    b = types_a.copy(); L.afgo_mp3_imdct_gr(b, ov2, 0, 0)      # all long
    c = types_a.copy(); ov3 = np.zeros(288, np.float32); L.afgo_mp3_imdct_gr(c, ov3, 2, 0)   # all short
    assert (a[:36] == b[:36]).all() and (a[36:] == c[36:]).all()


def test_antialias_is_a_rotation_and_change_sign():
    x = rng.standard_normal(576).astype(np.float32)
    y = x.copy(); L.afgo_mp3_antialias(y, 31)
    assert abs(float((y.astype(np.float64) ** 2).sum() / (x.astype(np.float64) ** 2).sum()) - 1) < 1e-6   # cs^2+ca^2 = 1
    z = x.copy(); L.afgo_mp3_antialias(z, 0)
    assert (z == x).all()
    s = x.copy(); L.afgo_mp3_change_sign(s)
    sign = np.ones((32, 18)); sign[1::2, 1::2] = -1
    assert (s.reshape(32, 18) == x.reshape(32, 18) * sign).all()


def test_dct2_matches_direct_sum():
    g = rng.standard_normal(576).astype(np.float32); g0 = g.copy()
    L.afgo_mp3_dct2(g, 18)
    kk = np.arange(32)[:, None]; nn = np.arange(32)[None, :]
    want = np.cos(np.pi / 32 * (nn + 0.5) * kk) @ g0.reshape(32, 18).astype(np.float64)
    assert np.abs(g.reshape(32, 18) - want).max() < 3e-5


def impulse_response(k, slot=0, nch=1, ch=0):
    qmf = np.zeros(960, np.float32); out = []
    for g in range(2):
        gr = np.zeros(1152, np.float32)
        if g == 0:
            gr[ch * 576 + k * 18 + slot] = 32768.0
        pcm = np.zeros(576 * nch, np.float32); lins = np.zeros((18 + 15) * 64, np.float32)
        L.afgo_mp3_synth_granule(qmf, gr, 18, nch, pcm, lins)
        out.append(pcm.reshape(576, nch)[:, ch].copy())
    return np.concatenate(out).astype(np.float64)


def test_polyphase_is_a_cosine_modulated_prototype():
    """ISO synthesis bank: h_k[n] = p[n] cos((2k+1)(n+16) pi/64); p must not depend on k."""
    n = np.arange(1152)
    h0 = impulse_response(0)
    assert np.abs(h0[512:]).max() == 0 and np.abs(h0[:512]).max() > 2.0     # 512-tap support, peak tap 75038/32768
    c0 = np.cos((n + 16) * np.pi / 64)
    for k in (1, 5, 17, 31):
        hk = impulse_response(k); ck = np.cos((2 * k + 1) * (n + 16) * np.pi / 64)
        m = (np.abs(c0) > 0.2) & (np.abs(ck) > 0.2) & (n < 512)
        assert np.abs(hk[m] / ck[m] - h0[m] / c0[m]).max() < 0.05


def test_polyphase_linearity_shift_invariance_and_stereo():
    h = impulse_response(7, slot=0); hs = impulse_response(7, slot=5)
    assert np.abs(hs[5 * 32:5 * 32 + 512] - h[:512]).max() < 1e-6            # 32-sample shift per slot
    l = impulse_response(7, nch=2, ch=0); r = impulse_response(7, nch=2, ch=1)
    assert (l == h).all() and (r == h).all()                                   # channels are independent copies


def test_batch_driver_state_and_mono_quirk():
    granules = [6, 4]; channels = [2, 1]
    coef = rng.standard_normal(576 * (12 + 4)).astype(np.float32)
    flags = np.full(16, oraclelib.mp3_flags(), np.uint32)
    pcm, st = oraclelib.mp3_transform(granules, channels, coef, flags, want_state=True)
    assert np.isfinite(pcm).all() and st.shape == (2, oraclelib.MP3_STATE_FLOATS)
    # zero spectrum -> zero PCM, zero state
    z, zst = oraclelib.mp3_transform(granules, channels, np.zeros_like(coef), flags, want_state=True)
    assert (z == 0).all() and (zst == 0).all()
