"""A FLAC *bitstream writer* written from the FLAC format description (RFC 9639 layout), the other half of
flac_ref_encoder.py: it serialises that encoder model's records into a native .flac file -- metadata
blocks, frame headers with CRC-8, CONSTANT / VERBATIM / FIXED / LPC subframes, partitioned Rice
residuals, frame CRC-16 -- so the host front-end (afg_flac_parse) can be checked against records it has
never seen, and whole files can be decoded end to end.  Test infrastructure only."""
import numpy as np

import flac_ref_encoder as enc

SR_TABLE = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9,
            48000: 10, 96000: 11}
BPS_TABLE = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6}
BS_TABLE = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12,
            8192: 13, 16384: 14, 32768: 15}


def crc8(data):
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xff if c & 0x80 else (c << 1) & 0xff
    return c


def crc16(data):
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xffff if c & 0x8000 else (c << 1) & 0xffff
    return c


class Bits:
    def __init__(self):
        self.parts = []

    def u(self, value, n):
        if n:
            self.parts.append(format(int(value) & ((1 << n) - 1), "0%db" % n))

    def s(self, value, n):
        self.u(int(value) & ((1 << n) - 1), n)

    def unary(self, zeros):
        self.parts.append("0" * int(zeros) + "1")

    def raw(self, bitstring):
        self.parts.append(bitstring)

    def nbits(self):
        return sum(len(p) for p in self.parts)

    def align(self):
        pad = (-self.nbits()) % 8
        if pad:
            self.parts.append("0" * pad)

    def tobytes(self):
        s = "".join(self.parts)
        assert len(s) % 8 == 0
        return int(s, 2).to_bytes(len(s) // 8, "big") if s else b""


def utf8_number(v):
    if v < 0x80:
        return bytes([v])
    n = 2
    while v >= 1 << (5 * n + 1):       # n bytes carry (7-n) + 6*(n-1) = 5n+1 bits
        n += 1
    out = [0] * n
    for i in range(n - 1, 0, -1):
        out[i] = 0x80 | (v & 0x3f)
        v >>= 6
    out[0] = ((0xff << (8 - n)) & 0xff) | v
    return bytes(out)


def rice_bits(res, k):
    u = (res.astype(np.int64) << 1) ^ (res.astype(np.int64) >> 63)      # zig-zag
    q = u >> k
    return int(q.sum()) + len(res) * (1 + k)


def rice_partition(bits, res, k):
    u = (res.astype(np.int64) << 1) ^ (res.astype(np.int64) >> 63)
    fmt = "0%db" % k
    mask = (1 << k) - 1
    out = []
    for v in u.tolist():
        out.append("0" * (v >> k) + "1" + (format(v & mask, fmt) if k else ""))
    bits.raw("".join(out))


def write_residual(bits, res, order, block_size, rice2=False, max_part_order=3, escape_partition=None, escape_bits=0):
    """Partitioned Rice (method 0 / 1).  escape_partition: index of a partition written unencoded with
    escape_bits per sample (only for negative tests: the reference does not decode it)."""
    part_order = 0
    for po in range(max_part_order, -1, -1):
        if block_size % (1 << po) == 0 and (block_size >> po) > order:
            part_order = po
            break
    bits.u(1 if rice2 else 0, 2)
    bits.u(part_order, 4)
    pbits, kmax = (5, 30) if rice2 else (4, 14)
    pos = order
    for part in range(1 << part_order):
        count = (block_size >> part_order) - (order if part == 0 else 0)
        seg = res[pos:pos + count]
        pos += count
        if escape_partition == part:
            bits.u((1 << pbits) - 1, pbits)
            bits.u(escape_bits, 5)
            for v in seg.tolist():
                bits.s(v, escape_bits)
            continue
        best = min(range(kmax + 1), key=lambda k: rice_bits(seg, k)) if len(seg) else 0
        bits.u(best, pbits)
        rice_partition(bits, seg, best)
    assert pos == block_size


def write_subframe(bits, sf, plane, sbps, kind=None, **kw):
    """plane: the residual plane of this subframe (warm-up samples first).  kind: None = pick from the record."""
    order, bs = int(sf["order"]), len(plane)
    coef = sf["coef"][:order].tolist()
    w = int(sf["wasted"])
    if kind is None:
        if order == 0 and bs and np.all(plane == plane[0]):
            kind = "constant"
        elif order == 0:
            kind = "verbatim" if (int(plane[0]) & 1) else "fixed"
        elif order <= 4 and coef == enc.FIXED[order] and int(sf["shift"]) == 0:
            kind = "fixed"
        else:
            kind = "lpc"
    code = {"constant": 0, "verbatim": 1, "fixed": 8 | order, "lpc": 0x20 | (order - 1)}[kind]
    bits.u(0, 1)
    bits.u(code, 6)
    bits.u(1 if w else 0, 1)
    if w:
        bits.unary(w - 1)
    if kind == "constant":
        bits.s(plane[0], sbps)
    elif kind == "verbatim":
        for v in plane.tolist():
            bits.s(v, sbps)
    else:
        for v in plane[:order].tolist():
            bits.s(v, sbps)
        if kind == "lpc":
            prec = kw.get("precision", 12)
            bits.u(prec - 1, 4)
            bits.s(int(sf["shift"]), 5)
            for c in coef:
                bits.s(c, prec)
        write_residual(bits, plane, order, bs, rice2=kw.get("rice2", False),
                       escape_partition=kw.get("escape_partition"), escape_bits=kw.get("escape_bits", 0))


def write_frame(fr, subframes, res, number, sample_rate, stream_bps, variable=False, header_bps=True, **kw):
    bs, C, asg = int(fr["block_size"]), int(fr["channels"]), int(fr["assignment"])
    bits = Bits()
    bits.u(0x3FFE, 14)
    bits.u(0, 1)
    bits.u(1 if variable else 0, 1)
    bs_code = BS_TABLE.get(bs, 6 if bs <= 256 else 7)
    sr_code = SR_TABLE.get(sample_rate)
    if sr_code is None:
        sr_code = 12 if sample_rate % 1000 == 0 and sample_rate < 256000 else (13 if sample_rate < 65536 else 14)
    bits.u(bs_code, 4)
    bits.u(sr_code, 4)
    bits.u(asg if asg >= 8 else C - 1, 4)
    bits.u(BPS_TABLE.get(stream_bps, 0) if header_bps else 0, 3)
    bits.u(0, 1)
    bits.raw("".join(format(b, "08b") for b in utf8_number(number)))
    if bs_code == 6:
        bits.u(bs - 1, 8)
    elif bs_code == 7:
        bits.u(bs - 1, 16)
    if sr_code == 12:
        bits.u(sample_rate // 1000, 8)
    elif sr_code == 13:
        bits.u(sample_rate, 16)
    elif sr_code == 14:
        bits.u(sample_rate // 10, 16)
    bits.u(crc8(bits.tobytes()), 8)
    base, sfi = int(fr["in_off"]), int(fr["sf_index"])
    for c in range(C):
        sf = subframes[sfi + c]
        sbps = stream_bps - int(sf["wasted"])
        if (asg in (enc.LEFT_SIDE, enc.MID_SIDE) and c == 1) or (asg == enc.RIGHT_SIDE and c == 0):
            sbps += 1
        write_subframe(bits, sf, res[base + c * bs: base + (c + 1) * bs], sbps, **kw)
    bits.align()
    body = bits.tobytes()
    return body + crc16(body).to_bytes(2, "big")


def metadata_block(kind, payload, last):
    return bytes([(0x80 if last else 0) | kind]) + len(payload).to_bytes(3, "big") + payload


def streaminfo(sample_rate, channels, bps, total_samples, min_bs, max_bs):
    b = Bits()
    b.u(min_bs, 16)
    b.u(max_bs, 16)
    b.u(0, 24)
    b.u(0, 24)
    b.u(sample_rate, 20)
    b.u(channels - 1, 3)
    b.u(bps - 1, 5)
    b.u(total_samples, 36)
    b.u(0, 128)                     # MD5 left unset
    return b.tobytes()


def write_file(frames, subframes, res, sample_rate, bps, total_samples=None, extra_metadata=(), variable=False, **kw):
    """Serialise the encoder model's records as a native FLAC file."""
    C = int(frames[0]["channels"])
    sizes = [int(f["block_size"]) for f in frames]
    if total_samples is None:
        total_samples = sum(sizes)
    blocks = [(0, streaminfo(sample_rate, C, bps, total_samples, min(sizes), max(sizes)))] + list(extra_metadata)
    out = [b"fLaC"]
    for i, (kind, payload) in enumerate(blocks):
        out.append(metadata_block(kind, payload, i == len(blocks) - 1))
    pos = 0
    for i, fr in enumerate(frames):
        out.append(write_frame(fr, subframes, res, pos if variable else i, sample_rate, bps, variable=variable, **kw))
        pos += sizes[i]
    return b"".join(out)


def encode_file(pcm, bps, block_size, sample_rate=44100, **kw):
    """pcm [frames, channels] ints -> (file bytes, records) using the encoder model's choices."""
    enc_kw = {k: kw.pop(k) for k in ("orders", "assignments", "use_fixed_every", "seed") if k in kw}
    frames, subframes, res, total = enc.encode(pcm, bps, block_size, **enc_kw)
    return write_file(frames, subframes, res, sample_rate, bps, **kw), (frames, subframes, res, total)
