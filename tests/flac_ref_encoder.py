"""A small FLAC *encoder model* written from the FLAC format description (not from the reference):
inter-channel decorrelation, wasted-bits detection, quantised LPC / fixed predictors and exact
integer residuals.  It emits the transform-stage records (frames / subframes / residual planes) the
host parser would hand to the restore stage, so decode(encode(pcm)) == pcm pins the oracle's
prediction direction, shift semantics and decorrelation formulas independently of its own code."""
import numpy as np

import oraclelib

INDEPENDENT, LEFT_SIDE, RIGHT_SIDE, MID_SIDE = 0, 8, 9, 10
FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def lpc_coefficients(x, order, precision=12):
    """Autocorrelation + Levinson-Durbin, quantised the way libFLAC does (coef, shift)."""
    x = x.astype(np.float64) * np.hanning(len(x))
    r = np.array([np.dot(x[:len(x) - k], x[k:]) for k in range(order + 1)])
    if r[0] == 0:
        return np.zeros(order, np.int16), 0
    a = np.zeros(order + 1)
    a[0] = 1.0
    err = r[0]
    for i in range(1, order + 1):
        acc = r[i] + np.dot(a[1:i], r[i - 1:0:-1])
        k = -acc / err
        a[1:i + 1] = a[1:i + 1] + k * np.concatenate([a[i - 1:0:-1], [1.0]])
        err *= (1 - k * k)
        if err <= 0:
            break
    lpc = -a[1:]
    cmax = np.abs(lpc).max()
    if cmax == 0:
        return np.zeros(order, np.int16), 0
    shift = int(np.clip(precision - 2 - int(np.floor(np.log2(cmax))), 0, 15))
    q = np.clip(np.round(lpc * (1 << shift)), -(1 << (precision - 1)), (1 << (precision - 1)) - 1)
    return q.astype(np.int16), shift


def residual(s, coef, shift):
    """res[t] = s[t] - ((sum_k coef[k] * s[t-1-k]) >> shift), exact integers; warm-up kept verbatim."""
    s = s.astype(np.int64)
    order = len(coef)
    res = s.copy()
    for t in range(order, len(s)):
        acc = 0
        for k in range(order):
            acc += int(coef[k]) * int(s[t - 1 - k])
        res[t] = s[t] - (acc >> shift)
    return res


def wasted_bits(s):
    nz = s[s != 0]
    if nz.size == 0:
        return 0
    k = 0
    while np.all((nz >> k) & 1 == 0):
        k += 1
    return k


def encode(pcm, bps, block_size, orders=(8, 12), assignments=(MID_SIDE, LEFT_SIDE, RIGHT_SIDE, INDEPENDENT),
           use_fixed_every=5, seed=0):
    """pcm: int array [frames, channels] within bps bits.  Returns (frames, subframes, res, total)."""
    rng = np.random.default_rng(seed)
    n, C = pcm.shape
    nfr = (n + block_size - 1) // block_size
    frames = np.zeros(nfr, oraclelib.FLAC_FRAME_DTYPE)
    subframes = np.zeros(nfr * C, oraclelib.FLAC_SUBFRAME_DTYPE)
    planes = []
    in_off = out_off = 0
    for f in range(nfr):
        blk = pcm[f * block_size:(f + 1) * block_size].astype(np.int64)
        bs = len(blk)
        asg = assignments[f % len(assignments)] if C == 2 else INDEPENDENT
        if asg == MID_SIDE:
            chans = [(blk[:, 0] + blk[:, 1]) >> 1, blk[:, 0] - blk[:, 1]]
            extra = [0, 1]
        elif asg == LEFT_SIDE:
            chans = [blk[:, 0], blk[:, 0] - blk[:, 1]]
            extra = [0, 1]
        elif asg == RIGHT_SIDE:
            chans = [blk[:, 0] - blk[:, 1], blk[:, 1]]
            extra = [1, 0]
        else:
            chans = [blk[:, c] for c in range(C)]
            extra = [0] * C
        frames[f] = (in_off, out_off, bs, f * C, C, asg, bps, 0, [0] * 4)
        for c, s in enumerate(chans):
            w = wasted_bits(s)
            s = s >> w
            sf_bps = bps + extra[c] - w
            if (f + c) % use_fixed_every == 0:
                order = min(int(rng.integers(0, 5)), bs)
                coef, shift = np.array(FIXED[order], np.int16), 0
            else:
                order = min(orders[(f + c) % len(orders)], bs)
                coef, shift = lpc_coefficients(s, order) if order else (np.zeros(0, np.int16), 0)
            sf = subframes[f * C + c]
            sf["coef"][:order] = coef
            sf["order"], sf["shift"], sf["wasted"], sf["use64"] = order, shift, w, int(sf_bps > 16)
            planes.append(residual(s, coef, shift).astype(np.int64))
        in_off += bs * C
        out_off += bs * C
    res = np.concatenate(planes)
    assert np.abs(res).max() < 2 ** 31
    return frames, subframes, res.astype(np.int32), out_off
