"""Hand-made inputs of the Vorbis coupling / floor stage (test infrastructure): steep and shallow segments both ways, y beyond
0..255 (a malformed stream's finalY: the table index wraps, the line arithmetic does not), repeated x, points beyond the
block, single-point curves, silent channels, chained coupling steps, special floats."""
import numpy as np

import afgpu


def hand_made(seed=5, trials=160):
    rng = np.random.default_rng(seed)
    packets, curves, points, steps, spec_n = [], [], [], [], 0
    for trial in range(trials):
        n2 = int(rng.choice([128, 256, 512, 1024, 4096]))
        ch = int(rng.integers(1, 5))
        nst = int(rng.integers(0, 4)) if ch > 1 else 0
        rec = np.zeros(1, afgpu.VORBIS_FLOOR_PACKET_DTYPE)
        rec["spec_off"], rec["n2"], rec["channels"] = spec_n, n2, ch
        rec["curve_index"], rec["step_off"], rec["n_steps"] = len(curves), len(steps), nst
        packets.append(rec)
        for _ in range(nst):
            m = int(rng.integers(0, ch))
            a = int((m + rng.integers(1, ch)) % ch)
            steps.append((m, a))
        for c in range(ch):
            kind = int(rng.integers(0, 6))
            if kind == 0:
                curves.append((len(points), 0))
                continue
            npt = 1 if kind == 1 else int(rng.integers(2, 60))
            hi = n2 * (4 if kind == 2 else 1)
            xs = np.sort(rng.integers(1, hi + 1, npt - 1)) if npt > 1 else np.zeros(0, np.int64)
            if kind == 3 and npt > 3:
                xs[1] = xs[2]                                                    # repeated x: the later point wins (lx != hx)
            yr = (-700, 900) if kind == 4 else (0, 256)
            ys = rng.integers(yr[0], yr[1], npt)
            if kind == 5:
                ys = np.where(rng.random(npt) < 0.5, 0, 255)                     # the steepest legal lines
            curves.append((len(points), npt))
            points.append((0, int(ys[0])))
            points.extend((int(x), int(y)) for x, y in zip(xs, ys[1:]))
        spec_n += n2 * ch
    spec = rng.standard_normal(spec_n).astype(np.float32) * np.float32(300)
    spec[rng.integers(0, spec_n, 400)] = 0.0
    spec[rng.integers(0, spec_n, 50)] = -0.0
    spec[rng.integers(0, spec_n, 20)] = np.inf
    spec[rng.integers(0, spec_n, 20)] = 1e-42                                     # denormal residues stay denormal arithmetic
    pk = np.concatenate(packets)
    cv = np.array(curves, dtype=afgpu.VORBIS_FLOOR_CURVE_DTYPE)
    pt = np.array(points, np.int32).reshape(-1, 2)
    st = np.array(steps, np.uint8).reshape(-1, 2)
    return pk, cv, pt, st, spec


def same_floats(got, want):
    """bit-equal, NaNs (inf * 0, inf - inf) equal as NaNs"""
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    bad = np.nonzero(~same)[0]
    assert len(bad) == 0, (len(bad), bad[:8], got[bad[:8]], want[bad[:8]])
