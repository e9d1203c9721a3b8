"""A Layer III *bitstream generator* written from the format description (ISO/IEC 11172-3 / 13818-3 frame layout),
test infrastructure only.  It draws random but valid side information and quantised spectra, codes them with the
code books (read from the generated oracle header, i.e. as data), packs the main data through a bit reservoir and
returns the file bytes together with what it encoded, so that

  * the product parser and the oracle can be compared on MPEG-1/2/2.5, mono / stereo / M-S / intensity, all block
    types (mixed included), every table, linbits escapes, scfsi, preflag, scalefac_scale, subblock gains; and
  * the requantiser can be checked against the textbook formula in float64
        xr = sign * |q|^(4/3) * 2^((global_gain - 210)/4) * 2^(-(1 + scalefac_scale)/2 * (sf + preflag*pretab))
    (short blocks: ... * 2^(-2 * subblock_gain[w])), which the records carry scaled by 1/2 (and 1/sqrt(2) with M/S).
"""
import os
import re

import numpy as np

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "mp3_front_tables.h")


def _load():
    text = open(HDR).read()

    def nums(name):
        m = re.search(name + r"(\[\d+\])+ = \{(.*?)\};", text, re.S)
        body = re.sub(r"/\*.*?\*/", "", m.group(2), flags=re.S)
        return [int(v, 0) for v in re.findall(r"-?0x[0-9a-fA-F]+|-?\d+", body)]
    t = {}
    t["long"] = np.array(nums("k_sfb_long")).reshape(8, 23)
    t["short"] = np.array(nums("k_sfb_short")).reshape(8, 40)
    t["mixed"] = np.array(nums("k_sfb_mixed")).reshape(8, 40)
    t["parts"] = np.array(nums("k_scf_partitions")).reshape(3, 28)
    t["scfc"] = nums("k_scfc_decode")
    t["mod"] = nums("k_scf_mod")
    t["preamp"] = nums("k_preamp")
    t["halfrate"] = np.array(nums("k_halfrate")).reshape(2, 3, 15)
    t["linbits"] = nums("k_linbits")
    t["book_of"] = nums("k_book_of_table")
    codes = np.array(nums("k_huff_codes")).reshape(-1, 4)
    first, count = nums("k_huff_first"), nums("k_huff_count")
    t["books"] = {b: {(int(x), int(y)): (int(ln), int(c)) for ln, c, x, y in codes[first[b]:first[b] + count[b]]}
                  for b in range(32) if count[b]}
    c1 = np.array(nums("k_count1_codes")).reshape(2, 16, 3)
    t["count1"] = [{int(v): (int(ln), int(c)) for ln, c, v in c1[k]} for k in range(2)]
    return t


T = _load()
PRETAB = [0] * 11 + T["preamp"] + [0]


class Bits:
    def __init__(self):
        self.s = []

    def put(self, v, n):
        if n:
            self.s.append(format(int(v) & ((1 << n) - 1), "0%db" % n))

    def n(self):
        return sum(len(x) for x in self.s)

    def bytes(self, pad_to=None):
        s = "".join(self.s)
        s += "0" * ((-len(s)) % 8)
        b = int(s, 2).to_bytes(len(s) // 8, "big") if s else b""
        if pad_to is not None:
            assert len(b) <= pad_to, (len(b), pad_to)
            b += bytes(pad_to - len(b))
        return b


def band_table(version, sr, block_type, mixed):
    idx = sr + {"mpeg1": 6, "mpeg2": 3, "mpeg25": 0}[version]      # minimp3.d:134-137
    idx -= idx != 0
    if block_type == 2:
        tab = T["mixed"][idx] if mixed else T["short"][idx]
    else:
        tab = T["long"][idx]
    w = [int(v) for v in tab]
    return w[:w.index(0)]


def scf_layout(version, g, ch, intensity):
    """(widths[4], counts[4]) of the scalefactor fields, ISO 2.4.2.7 (MPEG-1) / 13818-3 2.4.3.2 (LSF)."""
    n_short = 0 if g["block_type"] != 2 else (30 if g["mixed"] else 39)
    n_long = 22 if g["block_type"] != 2 else ((8 if version == "mpeg1" else 6) if g["mixed"] else 0)
    row = T["parts"][(1 if n_short else 0) + (0 if n_long else 1)]
    if version == "mpeg1":
        part = T["scfc"][g["scalefac_compress"]]
        return [part >> 2, part >> 2, part & 3, part & 3], [int(v) for v in row[:4]], n_long, n_short
    ist = 1 if (intensity and ch) else 0
    sfc = g["scalefac_compress"] >> ist
    k = ist * 12
    widths = [0] * 4
    while sfc >= 0:
        prod = 1
        for i in range(3, -1, -1):
            widths[i] = sfc // prod % T["mod"][k + i]
            prod *= T["mod"][k + i]
        sfc -= prod
        k += 4
    return widths, [int(v) for v in row[k:k + 4]], n_long, n_short


MIXED_P = 0.4                        # share of short blocks written as mixed blocks (an encoder hardly ever writes one)


def gen_granule(rng, version, sr, nch, ch, gr, mode_ms, intensity, budget_bits, prev_scf, prev_short=False, strict_after=None):
    """Random granule-channel.  Returns (side dict, main-data Bits, q[576] ints in coded order, iscf list)."""
    for attempt in range(20):
        g = {}
        bt = int(rng.choice([0, 0, 0, 1, 2, 2, 3]))
        if strict_after is not None:
            # what an encoder may write: long -> long | start, start -> short, short -> short | stop, stop -> long | start
            bt = int(rng.choice({0: [0, 0, 0, 1], 1: [2], 2: [2, 3], 3: [0, 0, 1]}[strict_after]))
        g["block_type"] = bt
        g["mixed"] = int(bt == 2 and rng.random() < MIXED_P)
        g["global_gain"] = int(rng.integers(120, 200))
        g["scalefac_scale"] = int(rng.integers(0, 2))
        g["count1_table"] = int(rng.integers(0, 2))
        g["subblock_gain"] = [int(v) for v in rng.integers(0, 8, 3)]
        if version == "mpeg1":
            g["scalefac_compress"] = int(rng.integers(0, 16))
            g["preflag"] = int(rng.integers(0, 2))
            # scfsi is void when either granule of the channel is short (the decoder masks it, minimp3.d:571)
            g["scfsi"] = int(rng.integers(0, 16)) if (gr == 1 and bt != 2 and not prev_short) else 0
        else:
            if intensity and ch:
                g["scalefac_compress"] = int(rng.integers(0, 512))
            else:
                g["scalefac_compress"] = int(rng.integers(0, 512))
            g["preflag"] = int(g["scalefac_compress"] >= 500) if not (intensity and ch) else int(g["scalefac_compress"] >= 500)
            g["scfsi"] = 0
        bands = band_table(version, sr, bt, g["mixed"])
        ends = np.cumsum(bands)
        if bt == 0:
            g["region"] = [int(rng.integers(0, 16)), int(rng.integers(0, 8))]
            if strict_after is not None:
                while g["region"][0] + g["region"][1] + 2 > 22:          # region boundaries inside the 22 long bands
                    g["region"] = [int(rng.integers(0, 16)), int(rng.integers(0, 8))]
            g["tables"] = [int(rng.choice([t for t in range(32) if t not in (4, 14)])) for _ in range(3)]
        else:
            g["region"] = [8 if (bt == 2 and not g["mixed"]) else 7, 255]
            g["tables"] = [int(rng.choice([t for t in range(32) if t not in (4, 14)])) for _ in range(2)] + [0]
        scale = rng.choice([0.02, 0.1, 0.3, 1.0])
        big_pairs = int(rng.integers(0, 289) * scale)
        quads = int(rng.integers(0, (576 - 2 * big_pairs) // 4 + 1) * rng.choice([0.0, 0.3, 1.0]))
        # region boundaries in lines
        def region_end(first_band, cnt):
            last = min(first_band + cnt, len(bands) - 1)
            return int(ends[last]), last + 1
        r0_end, nb = region_end(0, g["region"][0])
        r1_end, nb = region_end(nb, g["region"][1]) if nb < len(bands) else (576, nb)
        q = np.zeros(576, np.int64)
        bits = Bits()
        # ---- part 2: scalefactors ----
        widths, counts, n_long, n_short = scf_layout(version, g, ch, intensity)
        iscf = []
        scfsi = g["scfsi"] if version == "mpeg1" else -16
        pos = 0
        for i in range(4):
            cnt = counts[i]
            if not cnt:
                break
            if version == "mpeg1" and (scfsi & 8):
                iscf += list(prev_scf[pos:pos + cnt])
            else:
                for _ in range(cnt):
                    v = int(rng.integers(0, 1 << widths[i])) if widths[i] else 0
                    bits.put(v, widths[i])
                    iscf.append(v)
            pos += cnt
            scfsi = (scfsi * 2) if version == "mpeg1" else scfsi
        iscf += [0, 0, 0]
        iscf = iscf[:n_long + n_short] + [0] * max(0, n_long + n_short - len(iscf))
        # ---- part 3: big values ----
        ok = True
        for p in range(big_pairs):
            i = 2 * p
            reg = 0 if i < r0_end else (1 if i < r1_end else 2)
            t = g["tables"][reg]
            book = T["book_of"][t]
            lin = T["linbits"][t]
            if not book:
                continue                                  # table 0: zeros, no bits
            dim = max(x for x, _ in T["books"][book]) + 1
            pair = []
            for _ in range(2):
                r = rng.random()
                if r < 0.35:
                    v = 0
                elif lin and r > 0.93:
                    v = 15 + int(rng.integers(0, 1 << lin)) if rng.random() < 0.8 else 15 + (1 << lin) - 1
                else:
                    v = int(rng.integers(0, dim))
                pair.append((v, int(rng.integers(0, 2))))
            (x, sx), (y, sy) = pair
            ln, code = T["books"][book][(min(x, 15), min(y, 15))]
            bits.put(code, ln)
            for v, sgn in ((x, sx), (y, sy)):
                if lin and v >= 15:
                    bits.put(v - 15, lin)
                if v:
                    bits.put(sgn, 1)
            q[i] = -x if sx else x
            q[i + 1] = -y if sy else y
        # ---- count1 ----
        for k in range(quads):
            i = 2 * big_pairs + 4 * k
            vals = [int(rng.integers(0, 2)) for _ in range(4)]
            flags = (vals[0] << 3) | (vals[1] << 2) | (vals[2] << 1) | vals[3]
            ln, code = T["count1"][g["count1_table"]][flags]
            bits.put(code, ln)
            for j, v in enumerate(vals):
                if v:
                    sgn = int(rng.integers(0, 2))
                    bits.put(sgn, 1)
                    q[i + j] = -1 if sgn else 1
        if bits.n() <= min(budget_bits, 4095):
            g["big_values"] = big_pairs
            g["part2_3_length"] = bits.n()
            return g, bits, q, iscf, bands
    # could not fit: an empty granule
    g["big_values"] = 0
    g["part2_3_length"] = 0
    if version == "mpeg1":
        g["scalefac_compress"] = 0
        g["scfsi"] = 0
    else:
        g["scalefac_compress"] = 0
        g["preflag"] = 0
    widths, counts, n_long, n_short = scf_layout(version, g, ch, intensity)
    return g, Bits(), np.zeros(576, np.int64), [0] * (n_long + n_short), band_table(version, sr, g["block_type"], g["mixed"])


def side_info_bits(version, nch, begin, grs):
    b = Bits()
    if version == "mpeg1":
        b.put(begin, 9)
        b.put(0, 5 if nch == 1 else 3)
        for ch in range(nch):
            b.put(grs[1][ch]["scfsi"], 4)
    else:
        b.put(begin, 8)
        b.put(0, 1 if nch == 1 else 2)
    for gr in grs:
        for g in gr:
            b.put(g["part2_3_length"], 12)
            b.put(g["big_values"], 9)
            b.put(g["global_gain"], 8)
            b.put(g["scalefac_compress"], 4 if version == "mpeg1" else 9)
            if g["block_type"]:
                b.put(1, 1)
                b.put(g["block_type"], 2)
                b.put(g["mixed"], 1)
                b.put(g["tables"][0], 5)
                b.put(g["tables"][1], 5)
                for s in g["subblock_gain"]:
                    b.put(s, 3)
            else:
                b.put(0, 1)
                for t in g["tables"]:
                    b.put(t, 5)
                b.put(g["region"][0], 4)
                b.put(g["region"][1], 3)
            if version == "mpeg1":
                b.put(g["preflag"], 1)
            b.put(g["scalefac_scale"], 1)
            b.put(g["count1_table"], 1)
    return b


def expected_lines(version, g, q, iscf, bands, ms):
    """float64 requantisation of coded-order lines (before stereo processing and reorder), records scale."""
    shift = g["scalefac_scale"] + 1
    out = np.zeros(576)
    n_long = 22 if g["block_type"] != 2 else ((8 if version == "mpeg1" else 6) if g["mixed"] else 0)
    eff = list(iscf)
    if g["block_type"] == 2:
        for i in range(n_long, len(eff)):
            eff[i] = (eff[i] + (g["subblock_gain"][(i - n_long) % 3] << (3 - shift))) & 255
    elif g["preflag"]:
        for i in range(10):
            eff[11 + i] = (eff[11 + i] + T["preamp"][i]) & 255
    gain_exp = g["global_gain"] - 4 - 210 - (2 if ms else 0)
    pos = 0
    for b, w in enumerate(bands):
        sc = 2.0 ** (gain_exp / 4.0) * 2.0 ** (-((eff[b] << shift) / 4.0)) if b < len(eff) else 0.0
        seg = q[pos:pos + w].astype(np.float64)
        out[pos:pos + w] = np.sign(seg) * np.abs(seg) ** (4.0 / 3.0) * sc
        pos += w
    return out


# decoded PCM rms / rms of the requantised lines (records scale) of these streams, measured with the oracle: the IMDCT +
# polyphase chain is energy-preserving up to this factor (x sqrt(2) with M/S, whose lines carry 1/sqrt(2))
PCM_PER_LINE_RMS = 33.94


def make_file(seed, n_frames=6, version="mpeg1", sr=0, mode="stereo", bitrate_index=9, id3=False, pcm_rms=0.05, strict=False):
    """mode: mono | stereo | ms | intensity | ms+intensity.  Returns (bytes, list of per-frame dicts).
    pcm_rms: every granule's global_gain is chosen so that the decoded signal has about this rms (full scale = 1.0), as
    an encoder's material does -- the code words are random, the level is not (None: global_gain random in [120, 200),
    which decodes to tens to hundreds of times full scale)."""
    rng = np.random.default_rng(seed)
    nch = 1 if mode == "mono" else 2
    ms = "ms" in mode
    intensity = "intensity" in mode
    vbits = {"mpeg1": 3, "mpeg2": 2, "mpeg25": 0}[version]
    hz = [44100, 48000, 32000][sr] >> (0 if version == "mpeg1" else 1) >> (1 if version == "mpeg25" else 0)
    kbps = 2 * int(T["halfrate"][1 if version == "mpeg1" else 0][0][bitrate_index])       # layer III row
    samples = 1152 if version == "mpeg1" else 576
    ngr = 2 if version == "mpeg1" else 1
    side_bytes = (17 if nch == 1 else 32) if version == "mpeg1" else (9 if nch == 1 else 17)
    max_begin = 511 if version == "mpeg1" else 255
    frames = []
    stream = bytearray()              # main data of all frames back to back
    slot_pos = 0                      # payload bytes available before the current frame
    main_pos = 0
    prev_scf = [[0] * 64, [0] * 64]
    was_short = [False, False]
    last_bt = [0, 0]                 # strict=True: block types follow the window-switching state machine, regions stay inside the band table
    for f in range(n_frames):
        pad = int(rng.integers(0, 2))
        frame_bytes = samples * kbps * 125 // hz + pad
        cap = frame_bytes - 4 - side_bytes
        begin = slot_pos - main_pos
        assert 0 <= begin <= max_begin
        budget = (begin + cap) * 8
        grs, md, meta = [], Bits(), []
        for gr in range(ngr):
            row = []
            for ch in range(nch):
                share = max(0, (budget - md.n()) // ((ngr - gr) * nch - ch) - 8)
                g, bits, q, iscf, bands = gen_granule(rng, version, sr, nch, ch, gr, ms, intensity, share, prev_scf[ch],
                                                      prev_short=(gr == 1 and was_short[ch]), strict_after=(last_bt[ch] if strict else None))
                was_short[ch] = g["block_type"] == 2
                last_bt[ch] = g["block_type"]
                if pcm_rms is not None:
                    unity = dict(g, global_gain=214 + (2 if ms else 0))                  # gain_exp == 0 in expected_lines
                    r = float(np.sqrt(np.mean(expected_lines(version, unity, q, iscf, bands, ms) ** 2)))
                    if r > 0.0:
                        want = pcm_rms * float(rng.uniform(0.6, 1.4)) / (PCM_PER_LINE_RMS * (2.0 ** 0.5 if ms else 1.0))
                        g["global_gain"] = int(np.clip(round(unity["global_gain"] + 4.0 * np.log2(want / r)), 0, 255))
                row.append(g)
                md.s += bits.s
                if version == "mpeg1":
                    prev_scf[ch] = list(iscf) + [0] * 64
                meta.append({"gr": gr, "ch": ch, "g": g, "q": q, "iscf": iscf, "bands": bands})
            grs.append(row)
        main = md.bytes()
        # keep the reservoir inside its field: stuffing bytes after this frame's main data
        nxt = begin + cap - len(main)
        assert nxt >= 0
        if nxt > max_begin:
            main += bytes(nxt - max_begin)
        mode_bits = 3 if nch == 1 else (1 if (ms or intensity) else 0)
        ext = (2 if ms else 0) | (1 if intensity else 0)
        hdr = bytes([0xff, 0xE0 | (vbits << 3) | (1 << 1) | 1, (bitrate_index << 4) | (sr << 2) | (pad << 1),
                     (mode_bits << 6) | (ext << 4)])
        side = side_info_bits(version, nch, begin, grs).bytes(side_bytes)
        frames.append({"hdr": hdr, "side": side, "cap": cap, "meta": meta, "begin": begin})
        stream += main
        main_pos += len(main)
        slot_pos += cap
    total_cap = sum(fr["cap"] for fr in frames)
    payload = bytes(stream[:total_cap]) + bytes(max(0, total_cap - len(stream)))
    out = bytearray()
    if id3:
        out += b"ID3\x03\x00\x00\x00\x00\x00\x10" + bytes(16)
    at = 0
    for fr in frames:
        out += fr["hdr"] + fr["side"] + payload[at:at + fr["cap"]]
        at += fr["cap"]
    return bytes(out), frames, {"nch": nch, "hz": hz, "ms": ms, "intensity": intensity, "version": version, "sr": sr}
