"""The arithmetic of the device floor stage without a device: draw_line in closed form -- after k steps the error
accumulator has overflowed floor(k * ady' / adx) times -- stated in numpy exactly as audio-formats_amd/csrc/vorbis_floor.hip
computes it, against the oracle's restatement of the reference loop (stb_vorbis2.d:1534-1563, :2255-2284, :2493-2523)."""
import numpy as np

import afgpu
import oraclelib
import vorbis_floor_cases as cases


def inverse_db_table():
    tab = np.zeros(256, np.float32)
    p = np.zeros(1, afgpu.VORBIS_FLOOR_PACKET_DTYPE)
    p["n2"], p["channels"] = 4, 1
    c = np.array([(0, 1)], dtype=afgpu.VORBIS_FLOOR_CURVE_DTYPE)
    for y in range(256):                     # a one-point curve multiplies every bin by table[y]
        tab[y] = oraclelib.vorbis_floor(p, c, np.array([[0, y]], np.int32), np.zeros((0, 2), np.uint8), np.ones(4, np.float32))[0]
    return tab


def kernel_statement(pk, cv, pt, st, spec, tab):
    out = spec.copy()
    for k in pk:
        n2, ch, base = int(k["n2"]), int(k["channels"]), int(k["spec_off"])
        for s in range(int(k["n_steps"])):
            m, a = (int(v) for v in st[int(k["step_off"]) + s])
            mv, av = out[base + m * n2: base + (m + 1) * n2].copy(), out[base + a * n2: base + (a + 1) * n2].copy()
            with np.errstate(all="ignore"):
                m2 = np.where(mv > 0, np.where(av > 0, mv, mv + av), np.where(av > 0, mv, mv - av))
                a2 = np.where(mv > 0, np.where(av > 0, mv - av, mv), np.where(av > 0, mv + av, mv))
            out[base + m * n2: base + (m + 1) * n2], out[base + a * n2: base + (a + 1) * n2] = m2, a2
        for c in range(ch):
            cvr, t0 = cv[int(k["curve_index"]) + c], base + c * n2
            if cvr["n_points"] == 0:
                out[t0:t0 + n2] = 0
                continue
            P = pt[int(cvr["point_off"]):int(cvr["point_off"]) + int(cvr["n_points"])].astype(np.int64)
            j = np.arange(n2)
            seg = np.searchsorted(P[:, 0], j, side="right") - 1          # the last point with x <= j
            last = seg == len(P) - 1
            nxt = np.minimum(seg + 1, len(P) - 1)
            x0, y0, x1, y1 = P[seg, 0], P[seg, 1], P[nxt, 0], P[nxt, 1]
            dy, adx = y1 - y0, np.where(last, 1, x1 - x0)
            b = np.sign(dy) * (np.abs(dy) // adx)                        # C division truncates
            sy = np.where(dy < 0, b - 1, b + 1)
            ady = np.abs(dy) - np.abs(b) * adx
            kk = j - x0
            y = np.where(last, y0, y0 + kk * b + (sy - b) * ((kk * ady) // adx))
            with np.errstate(all="ignore"):
                out[t0:t0 + n2] = out[t0:t0 + n2] * tab[y & 255]
    return out


def test_table_is_the_specifications():
    tab = inverse_db_table()
    assert tab[255] == np.float32(1.0) and np.all(np.diff(tab) > 0) and abs(float(tab[0]) - 1.0649863e-07) < 1e-13


def test_closed_form_equals_the_reference_loop():
    tab = inverse_db_table()
    for seed in (5, 6, 7):
        pk, cv, pt, st, spec = cases.hand_made(seed, trials=80)
        cases.same_floats(kernel_statement(pk, cv, pt, st, spec, tab), oraclelib.vorbis_floor(pk, cv, pt, st, spec))
