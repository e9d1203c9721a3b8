"""An Ogg Opus *stream generator* written from the format descriptions (RFC 6716 section 3 packet framing, RFC 7845
Ogg encapsulation, RFC 3533 pages), test infrastructure only.  There is no Opus encoder in this image, so the audio
frames carry random bytes: a range decoder fed uniform bits emits symbols distributed the way its models expect,
i.e. every payload decodes to a statistically ordinary CELT frame, and the product parser and the oracle must agree
bit for bit on it -- silence flags, post-filter parameters, transients, band boosts, skipping, intensity and dual
stereo, every split / fold / anti-collapse path the bytes happen to reach."""
import numpy as np

from vorbis_bitstream import lacing, ogg_crc, page  # noqa: F401

CELT_FRAME_SIZES = (120, 240, 480, 960)


def toc(config, stereo, code):
    return bytes([(config << 3) | (int(bool(stereo)) << 2) | code])


def frame_len_bytes(n):
    """RFC 6716 3.2.1: one or two byte frame length"""
    if n < 252:
        return bytes([n])
    first = 252 + (n & 3)
    return bytes([first, (n - first) // 4])


def packet(rng, config, stereo, code, sizes=None, pad=0, vbr=None, count=None):
    """One Opus packet.  sizes: byte length of each frame (drawn if None)."""
    def draw():
        return int(rng.choice([0, 1, 2, 3, 8, 20, 40, 80, 120, 160, 250, 400, 700, 1275])) if rng.random() < 0.3 \
            else int(rng.integers(10, 260))
    if code == 0:
        n = draw() if sizes is None else sizes[0]
        return toc(config, stereo, 0) + rng.bytes(n)
    if code == 1:
        n = draw() if sizes is None else sizes[0]
        return toc(config, stereo, 1) + rng.bytes(2 * n)
    if code == 2:
        a, b = (draw(), draw()) if sizes is None else sizes
        return toc(config, stereo, 2) + frame_len_bytes(a) + rng.bytes(a + b)
    count = int(rng.integers(1, 5)) if count is None else count
    vbr = bool(rng.integers(0, 2)) if vbr is None else vbr
    head = toc(config, stereo, 3) + bytes([count | (0x40 if pad else 0) | (0x80 if vbr else 0)])
    if pad:
        p, enc = pad, b""
        while p >= 255:
            enc += b"\xff"
            p -= 254
        head += enc + bytes([p])
    if vbr:
        ss = [draw() for _ in range(count)] if sizes is None else list(sizes)
        body = b"".join(frame_len_bytes(x) for x in ss[:-1]) + rng.bytes(sum(ss))
    else:
        n = draw() if sizes is None else sizes[0]
        body = rng.bytes(n * count)
    return head + body + bytes(pad)


def packet_frames(pkt):
    """(frame_count, frame_size in samples) of a packet this module wrote"""
    config, code = pkt[0] >> 3, pkt[0] & 3
    size = CELT_FRAME_SIZES[config & 3] if config >= 16 else (480 << (config & 1) if config >= 12 else
                                                              max(480, 960 * (config & 3)))
    return ((1, 2, 2)[code] if code < 3 else (pkt[1] & 0x3f if len(pkt) > 1 else 0)), size


def opus_head(channels, preskip=312, gain=0, rate=48000, map_type=0, version=1, extra=b""):
    return (b"OpusHead" + bytes([version, channels]) + int(preskip).to_bytes(2, "little") +
            int(rate).to_bytes(4, "little") + int(gain & 0xffff).to_bytes(2, "little") + bytes([map_type]) + extra)


def opus_tags(vendor=b"afg test", comments=()):
    out = b"OpusTags" + len(vendor).to_bytes(4, "little") + vendor + len(comments).to_bytes(4, "little")
    for c in comments:
        out += len(c).to_bytes(4, "little") + c
    return out


def ogg_opus(packets, channels, preskip=312, gain=0, comments=(), serial=0x4f505553, packets_per_page=None, rng=None,
             trim=0, head=None, tags=None, first_granule=None, corrupt_page=None, bos=True):
    """Mux: OpusHead page (BOS), OpusTags page(s), audio pages.  A page's granule position is the sample count up to
    the end of the last packet that finishes on it (RFC 7845 section 4); the last page has it lowered by `trim`.
    first_granule overrides the first audio page's position, corrupt_page spoils the checksum of that page."""
    h = head if head is not None else opus_head(channels, preskip, gain)
    pages = [page(lacing(len(h)), h, 0x02 if bos else 0, 0, serial, 0)]
    t = tags if tags is not None else opus_tags(comments=comments)
    seq, segs, pos = 1, lacing(len(t)), 0
    while segs:                                          # the tags packet may span pages
        take, segs = segs[:255], segs[255:]
        nbytes = sum(take)
        pages.append(page(take, t[pos:pos + nbytes], 0x01 if pos else 0, 0xffffffffffffffff if segs else 0, serial, seq))
        pos += nbytes
        seq += 1
    gran, i, n, first = 0, 0, len(packets), True
    while i < n:
        k = packets_per_page if packets_per_page else (int(rng.integers(1, 9)) if rng is not None else 4)
        group, segs = [], []
        while i < n and len(group) < k and len(segs) + len(lacing(len(packets[i]))) <= 255:
            group.append(packets[i])
            segs += lacing(len(packets[i]))
            c, fs = packet_frames(packets[i])
            gran += c * fs
            i += 1
        last = i >= n
        g = gran - trim if last else gran
        if first and first_granule is not None:
            g = first_granule
        first = False
        pages.append(page(segs, b"".join(group), 0x04 if last else 0, g, serial, seq))
        seq += 1
    if corrupt_page is not None:
        p = bytearray(pages[corrupt_page])
        p[-1] ^= 0x55
        pages[corrupt_page] = bytes(p)
    return b"".join(pages)


def level_comment(packets, channels, preskip=312, pcm_rms=0.05):
    """The R128_TRACK_GAIN comment (Q7.8 dB, dopus.d:8011-8059) that brings the decode of these packets to about pcm_rms of
    full scale.  The payloads are random range-coder input, so their band energies -- and the level of the decode -- are
    anything; an encoder's file sits inside full scale.  Measured with the oracle's decode of the same packets ahead of
    the gain and the int16 conversion (this is test infrastructure); b"" if they do not decode."""
    import oraclelib
    rec = oraclelib.opus_decode_file(ogg_opus(packets, channels, preskip))
    if isinstance(rec, int) or rec.get("error") or not rec["pcm_frames"]:
        return b""
    base, recs = oraclelib.opus_channel_records(rec)
    pcm = oraclelib.celt_transform(base, recs, rec["coeffs"], rec["pcm_frames"] * rec["channels"]).astype(np.float64)
    rms = float(np.sqrt(np.mean(pcm ** 2)))
    if not np.isfinite(rms) or rms <= 0.0:
        return b""
    return b"R128_TRACK_GAIN=%d" % int(np.clip(round(5120.0 * np.log10(pcm_rms / rms)), -32768, 32767))


def random_celt_file(rng, channels, n_packets, preskip=312, gain=0, comments=(), configs=None, codes=None, mixed_stereo=True,
                     trim=None, pcm_rms=None):
    """A file of CELT-only packets (TOC configurations 16..31) with random framing codes and payloads.
    pcm_rms: add the track gain that brings the decode to about this level (level_comment)."""
    pkts = []
    for _ in range(n_packets):
        config = int(rng.choice(configs)) if configs is not None else int(rng.integers(16, 32))
        stereo = bool(rng.integers(0, 2)) if mixed_stereo else channels == 2
        code = int(rng.choice(codes)) if codes is not None else int(rng.choice([0, 0, 0, 1, 2, 3]))
        size = CELT_FRAME_SIZES[config & 3]
        limit = 2880 if channels == 2 else 5760          # what the reference's frame buffer holds per channel
        if code == 3:
            count = int(rng.integers(1, max(2, min(6, limit // size) + 1)))
            pad = int(rng.choice([0, 0, 1, 7, 254, 255, 300])) if rng.random() < 0.3 else 0
            pkts.append(packet(rng, config, stereo, 3, pad=pad, count=count))
        else:
            if code and 2 * size > limit:
                code = 0
            pkts.append(packet(rng, config, stereo, code))
    total = sum(c * fs for c, fs in map(packet_frames, pkts))
    if trim is None:
        trim = int(rng.integers(0, 400)) if total > preskip + 400 else 0
    if total - trim < preskip:
        trim = 0
        preskip = min(preskip, total)
    if pcm_rms is not None:
        c = level_comment(pkts, channels, preskip, pcm_rms)
        comments = tuple(comments) + ((c,) if c else ())
    return ogg_opus(pkts, channels, preskip, gain, comments, rng=rng, trim=trim), pkts
