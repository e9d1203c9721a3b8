"""Lane-level numpy model of vorbis_walk.hip's factorisation, one function per FFT size.

The kernel computes the n-sample inverse MDCT of a long block from ONE complex FFT of N = n/4 points held R = N/64 points
per lane, as register passes with LDS transposes between them:

    R = 4  (n = 1024):  4 x 4 x 4 x 4
    R = 8  (n = 2048):  8 x 8 x 8
    R = 16 (n = 4096): 16 x 4 x 16

This file restates the index algebra of those passes (which lane reads which LDS slot, which twiddle) on arrays shaped
[64 lanes][R], so that it can be checked against numpy's FFT on the CPU and so that the LDS addresses of every access can
be checked for bank conflicts (ds_read_b64 / ds_write_b64: two halves of 32 lanes, 64 banks of 4 bytes: the 8-byte slot
indices of a half must differ modulo 32).  tests/test_vorbis_walk_model.py runs it; the HIP code follows it line by line.
"""
import numpy as np

LANES = np.arange(64)


def group_of(lane):
    """Lanes 32..63 walk their point groups in descending order: lane ^ 32 holds group 63 - j."""
    return np.where(lane < 32, lane, 95 - lane)


class Lds:
    """A wavefront's transform area in complex slots; records the worst bank multiplicity of every access."""

    def __init__(self, slots):
        self.m = np.zeros(slots, np.complex128)
        self.worst = 1
        self.high = 0

    def _check(self, addr):
        addr = np.asarray(addr)
        assert addr.shape == (64,)
        self.high = max(self.high, int(addr.max()))
        for half in (addr[:32], addr[32:]):
            banks = half % 32
            self.worst = max(self.worst, int(np.bincount(banks, minlength=32).max()))

    def write(self, addr, val):
        self._check(addr)
        assert len(set(addr.tolist())) == 64, "two lanes write one slot"
        self.m[addr] = val

    def read(self, addr):
        self._check(addr)
        return self.m[addr].copy()


def dft(x, axis=-1):
    return np.fft.fft(x, axis=axis)


def w(n, e):
    return np.exp(-2j * np.pi * e / n)


# ---------------------------------------------------------------- R = 8 ------
def fft_r8(t, lds=None):
    """t[lane][r] = point group_of(lane) + 64 r  ->  Z[lane][s] = bin group_of(lane) + 64 s."""
    lds = lds or Lds(8 * 72)
    j = group_of(LANES)
    e = dft(t, axis=1) * w(512, j[:, None] * np.arange(8)[None, :])          # pass 1: over r -> k0
    for k in range(8):
        lds.write(j + 68 * k, e[:, k])
    n0, k0 = LANES >> 3, LANES & 7                                            # pass 2: lane (n0, k0) over n1 -> k1
    e = np.stack([lds.read(n0 + 8 * n1 + 68 * k0) for n1 in range(8)], axis=1)
    e = dft(e, axis=1) * w(64, n0[:, None] * np.arange(8)[None, :])
    for k1 in range(8):
        lds.write(k0 + 8 * k1 + 72 * n0, e[:, k1])
    e = np.stack([lds.read(j + 72 * n) for n in range(8)], axis=1)            # pass 3: lane j = k0 + 8 k1 over n0 -> k2
    return dft(e, axis=1), lds


# ---------------------------------------------------------------- R = 16 -----
def fft_r16(t, lds=None):
    """1024 points as 16 x 4 x 16: q = n3 + 16 n2 + 64 n1 -> k = k1 + 16 k2 + 64 k3."""
    lds = lds or Lds(16 * 72)
    j = group_of(LANES)
    e = dft(t, axis=1) * w(1024, j[:, None] * np.arange(16)[None, :])        # pass 1: over r = n1 -> k1
    for k1 in range(16):
        lds.write(j + 72 * k1, e[:, k1])
    # pass 2: lane (n3, g) = (lane >> 2, lane & 3) takes the four items k1 = g + 4 i, each a 4-point DFT over n2 -> k2
    n3, g = LANES >> 2, LANES & 3
    out = {}
    rd = [[lds.read(n3 + 16 * n2 + 72 * (g + 4 * i)) for n2 in range(4)] for i in range(4)]
    for i in range(4):
        x = dft(np.stack(rd[i], axis=1), axis=1) * w(64, n3[:, None] * np.arange(4)[None, :])
        for k2 in range(4):
            out[(i, k2)] = x[:, k2]
    for i in range(4):
        for k2 in range(4):
            lds.write((g + 4 * i) + 16 * k2 + 68 * n3, out[(i, k2)])
    e = np.stack([lds.read(j + 68 * n) for n in range(16)], axis=1)           # pass 3: lane j = k1 + 16 k2 over n3 -> k3
    return dft(e, axis=1), lds


# ---------------------------------------------------------------- R = 4 ------
def b_slot(n4, n3, kk):
    return n4 + 4 * n3 + 16 * kk + 4 * (kk >> 1)


def fft_r4(t, lds=None):
    """256 points as 4 x 4 x 4 x 4: q = n4 + 4 n3 + 16 n2 + 64 n1 -> k = k1 + 4 k2 + 16 k3 + 64 k4."""
    lds = lds or Lds(4 * 80)
    j = group_of(LANES)
    e = dft(t, axis=1) * w(256, j[:, None] * np.arange(4)[None, :])          # pass 1: over r = n1 -> k1
    for k1 in range(4):
        lds.write(j + 80 * k1, e[:, k1])
    m, k1 = LANES & 15, LANES >> 4                                            # pass 2: lane (m = n4 + 4 n3, k1) over n2 -> k2
    e = np.stack([lds.read(m + 16 * n2 + 80 * k1) for n2 in range(4)], axis=1)
    e = dft(e, axis=1) * w(64, m[:, None] * np.arange(4)[None, :])
    for k2 in range(4):
        lds.write(b_slot(m & 3, m >> 2, k1 + 4 * k2), e[:, k2])
    n4, kk = LANES & 3, LANES >> 2                                            # pass 3: lane (n4, kk = k1 + 4 k2) over n3 -> k3
    e = np.stack([lds.read(b_slot(n4, n3, kk)) for n3 in range(4)], axis=1)
    e = dft(e, axis=1) * w(16, n4[:, None] * np.arange(4)[None, :])
    for k3 in range(4):
        lds.write(kk + 16 * k3 + 72 * n4, e[:, k3])
    e = np.stack([lds.read(j + 72 * n) for n in range(4)], axis=1)            # pass 4: lane j = kk + 16 k3 over n4 -> k4
    return dft(e, axis=1), lds


FFT = {4: fft_r4, 8: fft_r8, 16: fft_r16}


# ---------------------------------------------------------------- the transform around the FFT
def spectrum_to_lanes(X, R):
    """xin[lane][r] = (X[2q], X[2q+1]) at q = group_of(lane) + 64 r, as the loads deliver it."""
    q = group_of(LANES)[:, None] + 64 * np.arange(R)[None, :]
    return X[2 * q], X[2 * q + 1]


def pretwiddle(ev, od, R):
    """t[q] = (X[2q] + i X[n/2-1-2q]) w[q]: the odd entry of point N-1-q sits in slot R-1-r of lane ^ 32."""
    n = 256 * R
    other = od[LANES ^ 32][:, ::-1]
    q = group_of(LANES)[:, None] + 64 * np.arange(R)[None, :]
    return (ev + 1j * other) * w(n, q + 0.125)


def posttwiddle(Z, R):
    n = 256 * R
    k = group_of(LANES)[:, None] + 64 * np.arange(R)[None, :]
    return Z * w(n, k + 0.125)


def u_of_c(c, R):
    """The DCT-IV u (n/2 values) from c[lane][s]: u[2k] = Re c[k], u[n/2-1-2k] = -Im c[k]."""
    n2 = 128 * R
    k = group_of(LANES)[:, None] + 64 * np.arange(R)[None, :]
    u = np.zeros(n2)
    u[2 * k] = c.real
    u[n2 - 1 - 2 * k] = -c.imag
    return u


def y_of_u(u, n):
    """stb_vorbis' inverse_mdct output (n samples) as the odd / even extension of u."""
    m = np.arange(n)
    y = np.empty(n)
    a, b = m < n // 4, (m >= n // 4) & (m < 3 * n // 4)
    y[a] = u[n // 4 + m[a]]
    y[b] = -u[3 * n // 4 - 1 - m[b]]
    y[~(a | b)] = -u[m[~(a | b)] - 3 * n // 4]
    return y


def imdct_reference(X, n):
    """y[m] = sum_k X[k] cos(pi/(2n) (2m + 1 + n/2)(2k + 1))   (SURVEY 8c)."""
    m = np.arange(n)[:, None]
    k = np.arange(n // 2)[None, :]
    return (np.cos(np.pi / (2 * n) * (2 * m + 1 + n // 2) * (2 * k + 1)) * X[None, :]).sum(axis=1)


def imdct_model(X, R):
    ev, od = spectrum_to_lanes(X, R)
    Z, lds = FFT[R](pretwiddle(ev, od, R))
    c = posttwiddle(Z, R)
    return y_of_u(u_of_c(c, R), 256 * R), c, lds


def tdac_frames(c_cur, cb, win, R):
    """Frames of a long block between two long blocks, written from the lanes' own registers.

    c_cur[lane][s]; cb[lane][2 i], cb[lane][2 i + 1] = u_prev[n/2-1-2k], u_prev[n/2-2-2k] for k = j + 64 (R/2 + i).
    Returns (out[n/2], new cb)."""
    n2, N, H = 128 * R, 64 * R, R // 2
    j = group_of(LANES)
    other = c_cur[LANES ^ 32]
    out = np.zeros(n2)
    new = np.zeros_like(cb)
    for i in range(H):
        a0 = c_cur[:, H + i].real                    # u[2k], k = j + 64 (H + i): frame j0 = 2 (j + 64 i)
        a1 = -other[:, H - 1 - i].imag               # u[2k + 1] = -Im c[N-1-k]
        b0, b1 = cb[:, 2 * i], cb[:, 2 * i + 1]
        j0 = 2 * (j + 64 * i)
        out[j0] = a0 * win[j0] - b0 * win[n2 - 1 - j0]
        out[j0 + 1] = a1 * win[j0 + 1] - b1 * win[n2 - 2 - j0]
        out[n2 - 1 - j0] = -a0 * win[n2 - 1 - j0] - b0 * win[j0]
        out[n2 - 2 - j0] = -a1 * win[n2 - 2 - j0] - b1 * win[j0 + 1]
        new[:, 2 * i] = -c_cur[:, H + i].imag        # u[n/2-1-2k]
        new[:, 2 * i + 1] = other[:, H - 1 - i].real  # u[2 (N-1-k)] = u[n/2-2-2k]
    return out, new


def vorbis_window(n):
    i = np.arange(n // 2)
    return np.sin(0.5 * np.pi * np.sin((i + 0.5) / (n // 2) * 0.5 * np.pi) ** 2)
