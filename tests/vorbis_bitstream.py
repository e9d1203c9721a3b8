"""An Ogg Vorbis I *stream generator* written from the format description (Vorbis I specification, RFC 3533 framing),
test infrastructure only.  It writes structurally valid identification / comment / set-up headers with random but
legal contents -- complete prefix-code books (ordered, dense and sparse length lists), lookup type 1 and 2 vector
books with and without sequence_p, floor 1 configurations, residues of type 0, 1 and 2, several submaps, channel
coupling, any channel count and block-size pair -- and audio packets whose payload after the mode bits is random:
with complete code books every bit string decodes to something, so the product parser and the oracle must agree bit
for bit on paths no real file in this image exercises."""
import numpy as np

_crc_table = []
for _i in range(256):
    _r = _i << 24
    for _ in range(8):
        _r = ((_r << 1) ^ 0x04c11db7) & 0xffffffff if _r & 0x80000000 else (_r << 1) & 0xffffffff
    _crc_table.append(_r)


def ogg_crc(data):
    crc = 0
    for b in data:
        crc = ((crc << 8) & 0xffffffff) ^ _crc_table[((crc >> 24) & 0xff) ^ b]
    return crc


class LsbBits:
    """Vorbis bit packing: values are written LSB first into successive bytes."""

    def __init__(self):
        self.acc = 0
        self.n = 0

    def put(self, v, bits):
        if bits:
            self.acc |= (int(v) & ((1 << bits) - 1)) << self.n
            self.n += bits

    def bytes(self):
        nbytes = (self.n + 7) // 8
        return self.acc.to_bytes(nbytes, "little") if nbytes else b""


def ilog(n):
    return int(n).bit_length() if n > 0 else 0


def page(packets_segments, body, flags, granule, serial, seq):
    hdr = bytearray(b"OggS\x00" + bytes([flags]) + int(granule & 0xffffffffffffffff).to_bytes(8, "little") +
                    serial.to_bytes(4, "little") + seq.to_bytes(4, "little") + bytes(4) + bytes([len(packets_segments)]) +
                    bytes(packets_segments))
    crc = ogg_crc(bytes(hdr) + body)
    hdr[22:26] = crc.to_bytes(4, "little")
    return bytes(hdr) + body


def lacing(n):
    return [255] * (n // 255) + [n % 255]


def random_lengths(rng, used, max_len=16):
    """code lengths of a complete prefix code with `used` words (random binary tree)"""
    if used == 1:
        return [1]                                   # a single used entry: the specification allows it
    leaves = [0]
    while len(leaves) < used:
        i = int(rng.integers(0, len(leaves)))
        if leaves[i] >= max_len:
            cand = [k for k, v in enumerate(leaves) if v < max_len]
            i = cand[int(rng.integers(0, len(cand)))]
        d = leaves.pop(i)
        leaves += [d + 1, d + 1]
    rng.shuffle(leaves)
    return [int(v) for v in leaves]


# development switches of the independent-decoder comparison (tests/golden/make_independent.py): names of generator features to leave out
SIMPLE = set()
LOOKUP1_ONLY = False                 # vector books of lookup type 1 only (FFmpeg's Vorbis decoder refuses type 2: tests/golden/make_independent.py)


def write_codebook(b, rng, entries, dim, kind, lookup, level_exp=0):
    """kind: 'ordered' | 'dense' | 'sparse'.  lookup: 0 | 1 | 2 (for 1, entries must be values**dim).
    level_exp: added to the exponents of the book's minimum and delta values -- every vector of the book times 2^level_exp."""
    b.put(0x564342, 24)
    b.put(dim, 16)
    b.put(entries, 24)
    if "dense_only" in SIMPLE:
        kind = "dense"
    if kind == "ordered":
        lens = sorted(random_lengths(rng, entries))
        b.put(1, 1)
        cur = 0
        length = lens[0]
        b.put(length - 1, 5)
        while cur < entries:
            cnt = sum(1 for v in lens[cur:] if v == length)
            b.put(cnt, ilog(entries - cur))
            cur += cnt
            length += 1
    else:
        b.put(0, 1)
        if kind == "sparse":
            used = max(2, int(entries * rng.uniform(0.05, 0.2)))       # < 1/4 used: stays sparse in the reference
            mask = np.zeros(entries, bool)
            mask[rng.choice(entries, used, replace=False)] = True
            lens = iter(random_lengths(rng, used))
            b.put(1, 1)
            for e in range(entries):
                b.put(int(mask[e]), 1)
                if mask[e]:
                    b.put(next(lens) - 1, 5)
        else:
            b.put(0, 1)
            for v in random_lengths(rng, entries):
                b.put(v - 1, 5)
    b.put(lookup, 4)
    if lookup:
        def pack_float(mant, exp, neg):              # specification 9.2.2: 21-bit mantissa, 10-bit biased exponent
            return (0x80000000 if neg else 0) | ((exp + 788) << 21) | mant
        b.put(pack_float(int(rng.integers(0, 1 << 12)), int(rng.integers(-16, -6)) + level_exp, int(rng.integers(0, 2))), 32)   # minimum
        b.put(pack_float(int(rng.integers(1, 1 << 10)), int(rng.integers(-14, -8)) + level_exp, 0), 32)                          # delta
        value_bits = int(rng.integers(1, 9))
        b.put(value_bits - 1, 4)
        b.put(int(rng.random() < 0.3 and "no_seq" not in SIMPLE), 1)            # sequence_p
        if lookup == 1:
            vals = round(entries ** (1.0 / dim))
            assert vals ** dim == entries
            n = vals
        else:
            n = entries * dim
        for _ in range(n):
            b.put(int(rng.integers(0, 1 << value_bits)), value_bits)


def make_file(seed, channels=2, bs=(256, 2048), n_packets=24, rate=44100, residue_types=(0, 1, 2), packet_bytes=(20, 400),
              force_long_only=False, break_windows_at=None, pcm_rms=0.05):
    """A random legal Ogg Vorbis file.  pcm_rms: the level of the decoded signal (full scale = 1.0): the packets are random
    bits, so the file is written twice -- once as drawn, decoded (by the oracle: this is test infrastructure), and again
    with every code book's value range moved by the power of two that brings the decode to about pcm_rms (the same draws:
    the residue vectors scale exactly, floor and transform are linear in them).  A draw that decodes to silence stays as
    it is (rms 0: callers that need signal check).  None: one pass, as drawn (rms anywhere from 0 to a thousand)."""
    args = (seed, channels, bs, n_packets, rate, residue_types, packet_bytes, force_long_only, break_windows_at)
    data = _make_file(*args, level_exp=0)
    if pcm_rms is None:
        return data
    import oraclelib
    rec = oraclelib.vorbis_decode_file(data)
    if rec is None:
        return data
    pcm = oraclelib.vorbis_file_pcm(rec).astype(np.float64)
    rms = float(np.sqrt(np.mean(pcm ** 2))) if pcm.size else 0.0
    if not np.isfinite(rms) or rms <= 0.0:
        return data
    k = int(np.clip(round(float(np.log2(pcm_rms / rms))), -200, 60))
    return _make_file(*args, level_exp=k) if k else data


def _make_file(seed, channels, bs, n_packets, rate, residue_types, packet_bytes, force_long_only, break_windows_at, level_exp):
    rng = np.random.default_rng(seed)
    serial = int(rng.integers(1, 1 << 31))
    log0, log1 = bs[0].bit_length() - 1, bs[1].bit_length() - 1
    ident = b"\x01vorbis" + (0).to_bytes(4, "little") + bytes([channels]) + rate.to_bytes(4, "little") + bytes(12) + \
        bytes([(log1 << 4) | log0, 1])
    comment = b"\x03vorbis" + (4).to_bytes(4, "little") + b"afgp" + (1).to_bytes(4, "little") + (7).to_bytes(4, "little") + \
        b"k=value" + b"\x01"

    # ---- set-up ----
    b = LsbBits()
    b.put(5, 8)
    for c in b"vorbis":
        b.put(c, 8)
    books = []                                       # (entries, dim, lookup)

    def add_book(entries, dim, kind, lookup):
        books.append((entries, dim, lookup, kind))
        return len(books) - 1
    # scalar books for floors (symbols are Y values / class selectors) and residue classification
    scalar = [add_book(int(rng.integers(2, 65)), 1, rng.choice(["ordered", "dense"]), 0) for _ in range(3)]
    scalar.append(add_book(int(rng.integers(40, 200)), 1, "sparse", 0))
    single = add_book(int(rng.integers(3, 9)), 1, "sparse" if rng.random() < 0.5 else "dense", 0) if False else None
    # vector books for residues
    vq = []
    for _ in range(5):
        if rng.random() < 0.5 or LOOKUP1_ONLY:
            dim = int(rng.choice([1, 2, 4]))
            vals = int(rng.integers(2, 5))
            vq.append(add_book(vals ** dim, dim, rng.choice(["ordered", "dense"]), 1))
        else:
            dim = int(rng.choice([1, 2, 4, 8]))
            vq.append(add_book(int(rng.integers(4, 40)), dim, rng.choice(["ordered", "dense", "sparse"]), 2))
    class_books = []
    b.put(len(books) + 2 - 1, 8)                     # two classification books are added below
    n_resid = 2
    resid_cfg = []
    for r in range(n_resid):
        classifications = int(rng.integers(2, 5))          # (a 1-entry book is an incomplete code: not generated)
        classwords = int(rng.integers(1, 4))
        class_books.append((classifications ** classwords, classwords))
        resid_cfg.append((classifications, classwords))
    all_books = list(books) + [(e, d, 0, "dense") for (e, d) in class_books]
    for (entries, dim, lookup, kind) in all_books:
        if kind == "sparse" and entries < 8:
            kind = "dense"
        write_codebook(b, rng, entries, dim, kind, lookup, level_exp)
    b.put(0, 6)                                      # time-domain transforms: one, value 0
    b.put(0, 16)
    # floors
    n_floors = 2
    b.put(n_floors - 1, 6)
    for _ in range(n_floors):
        b.put(1, 16)
        partitions = int(rng.integers(1, 6))
        classes = int(rng.integers(1, 4))
        if "simple_floor" in SIMPLE:
            classes = 1
        b.put(partitions, 5)
        plist = [int(rng.integers(0, classes)) for _ in range(partitions)]
        for c in plist:
            b.put(c, 4)
        cdims = []
        for c in range(max(plist) + 1):
            cdim = int(rng.integers(1, 5))
            sub = int(rng.integers(0, 3))
            if "simple_floor" in SIMPLE:
                sub = 0
            cdims.append(cdim)
            b.put(cdim - 1, 3)
            b.put(sub, 2)
            if sub:
                b.put(int(rng.choice(scalar)), 8)
            for _k in range(1 << sub):
                b.put(0 if (rng.random() < 0.2 and "simple_floor" not in SIMPLE) else int(rng.choice(scalar)) + 1, 8)     # 0 = no book (value 0)
        mult1 = int(rng.integers(0, 4))
        if "no_mult3" in SIMPLE and mult1 == 2:      # multiplier 3: range 86 in 7-bit words, i.e. ordinates past the range are writable
            mult1 = 1
        b.put(mult1, 2)                              # multiplier - 1
        rangebits = int(rng.integers(6, 10))
        b.put(rangebits, 4)
        count = sum(cdims[c] for c in plist)
        xs = rng.choice(np.arange(1, 1 << rangebits), count, replace=False)
        for x in xs:
            b.put(int(x), rangebits)
    # residues
    b.put(n_resid - 1, 6)
    for r in range(n_resid):
        rtype = int(residue_types[r % len(residue_types)])
        classifications, classwords = resid_cfg[r]
        part = int(rng.choice([2, 4, 8, 16, 32]))
        begin = int(rng.integers(0, 3)) * part
        end = begin + part * int(rng.integers(2, 40))
        if "res_end_small" in SIMPLE:                # inside the shortest spectrum (blocksize_0 / 2 values per channel)
            end = min(end, (bs[0] // 2 // part) * part)
            begin = min(begin, max(0, end - part))
        b.put(rtype, 16)
        b.put(begin, 24)
        b.put(end, 24)
        b.put(part - 1, 24)
        b.put(classifications - 1, 6)
        b.put(len(books) + r, 8)                     # classbook
        cascades = [int(rng.integers(0, 8)) | (int(rng.integers(0, 4)) << 3 if (rng.random() < 0.3 and "low_cascade" not in SIMPLE) else 0)
                    for _ in range(classifications)]
        for c in cascades:
            b.put(c & 7, 3)
            hi = c >> 3
            b.put(1 if hi else 0, 1)
            if hi:
                b.put(hi, 5)
        for c in cascades:
            for k in range(8):
                if c & (1 << k):
                    # type 0 needs part_size divisible by the book dimension
                    ok = [q for q in vq if part % books[q][1] == 0]
                    b.put(int(rng.choice(ok)), 8)
    # mappings
    n_maps = 2
    b.put(n_maps - 1, 6)
    for m in range(n_maps):
        b.put(0, 16)
        submaps = int(rng.integers(1, 3)) if (channels > 1 and "one_submap" not in SIMPLE) else 1
        b.put(1 if submaps > 1 else 0, 1)
        if submaps > 1:
            b.put(submaps - 1, 4)
        steps = int(rng.integers(0, min(channels, 3))) if (channels > 1 and "no_coupling" not in SIMPLE) else 0
        if "plain_coupling" in SIMPLE:
            steps = min(steps, 1)
        b.put(1 if steps else 0, 1)
        if steps:
            b.put(steps - 1, 8)
            for _ in range(steps):
                mag, ang = rng.choice(channels, 2, replace=False)
                if "plain_coupling" in SIMPLE:               # what an encoder writes for a stereo pair: one step, magnitude 0, angle 1
                    mag, ang = 0, 1
                b.put(int(mag), ilog(channels - 1))
                b.put(int(ang), ilog(channels - 1))
        b.put(0, 2)
        if submaps > 1:
            for _ in range(channels):
                b.put(int(rng.integers(0, submaps)), 4)
        for _ in range(submaps):
            b.put(0, 8)
            b.put(int(rng.integers(0, n_floors)), 8)
            b.put(int(rng.integers(0, n_resid)), 8)
    # modes
    modes = [(0, 0), (1, 1 % n_maps), (1, 0), (0, 1 % n_maps)]
    if force_long_only:
        modes = [(1, 0), (1, 1 % n_maps)]
    b.put(len(modes) - 1, 6)
    for flag, mp in modes:
        b.put(flag, 1)
        b.put(0, 16)
        b.put(0, 16)
        b.put(mp, 8)
    b.put(1, 1)                                      # framing
    setup = b.bytes()

    out = page([30], ident, 2, 0, serial, 0)
    hdr_body = comment + setup
    out += page(lacing(len(comment)) + lacing(len(setup)), hdr_body, 0, 0, serial, 1)

    # ---- audio packets: mode bits + random payload, a few packets per page, granule positions that add up ----
    pkts = []
    mode_seq = [int(rng.integers(0, len(modes))) for _ in range(n_packets)]
    if break_windows_at is not None:
        pass
    for k in range(n_packets):
        mi = mode_seq[k]
        pb = LsbBits()
        pb.put(0, 1)
        pb.put(mi, ilog(len(modes) - 1))
        is_long = modes[mi][0]
        if is_long:
            # window flags as an encoder writes them: does the neighbour use the long block size?
            prev_long = modes[mode_seq[k - 1]][0] if k else 1
            next_long = modes[mode_seq[k + 1]][0] if k + 1 < n_packets else 1
            if break_windows_at == k:
                prev_long ^= 1                              # deliberately inconsistent with the previous packet
            pb.put(prev_long, 1)
            pb.put(next_long, 1)
        nbytes = int(rng.integers(packet_bytes[0], packet_bytes[1]))
        for _ in range(nbytes):
            pb.put(int(rng.integers(0, 256)), 8)
        if rng.random() < 0.15 and "no_short_packets" not in SIMPLE:
            pkts.append(pb.bytes()[:int(rng.integers(1, 4))])             # a very short packet: end-of-packet paths
        else:
            pkts.append(pb.bytes())
    seq = 2
    i = 0
    gran = 0
    while i < len(pkts):
        take = int(rng.integers(1, 6))
        group = pkts[i:i + take]
        i += len(group)
        gran += sum(1024 for _ in group)
        segs = []
        for p in group:
            segs += lacing(len(p))
        last = i >= len(pkts)
        out += page(segs, b"".join(group), 4 if last else 0, gran if rng.random() < 0.9 else 0xffffffffffffffff, serial, seq)
        seq += 1
    return out
