"""MPEG audio Layer I / II *stream generator* written from the format description (ISO/IEC 11172-3 section 2.4, 13818-3),
test infrastructure only.

Two kinds of frames:
  * `layer1_frame`: a Layer I frame written field by field (allocation, scalefactor indices, sample codes) together
    with the subband samples the standard's requantisation formula gives for it,
        s = (code - (2^(nb-1) - 1)) * 2 / (2^nb - 1) * 2^(-index / 3)
    so that a decoder's numbers can be checked against the text of the standard;
  * `random_frame`: a valid header of any layer I / II configuration followed by random bits.  Every allocation index,
    scalefactor selection and sample code is legal to the reference's decoder, so the product parser and the oracle must
    agree bit for bit on whatever the bits happen to say (all allocation tables, grouped 3 / 5 / 9-level samples, joint
    stereo bounds, every bit rate / sampling rate / MPEG version)."""
import numpy as np

HALFRATE = {  # kbit/s / 2 by [mpeg1][layer 3, 2, 1][bit-rate index] (the standard's bit-rate tables)
    0: [[0, 4, 8, 12, 16, 20, 24, 28, 32, 40, 48, 56, 64, 72, 80], [0, 4, 8, 12, 16, 20, 24, 28, 32, 40, 48, 56, 64, 72, 80],
        [0, 16, 24, 28, 32, 40, 48, 56, 64, 72, 80, 88, 96, 112, 128]],
    1: [[0, 16, 20, 24, 28, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160], [0, 16, 24, 28, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192],
        [0, 16, 32, 48, 64, 80, 96, 112, 128, 144, 160, 176, 192, 208, 224]],
}
MODES = {"stereo": 0, "joint": 1, "dual": 2, "mono": 3}


class MsbBits:
    def __init__(self):
        self.v = 0
        self.n = 0

    def put(self, value, bits):
        self.v = (self.v << bits) | (int(value) & ((1 << bits) - 1))
        self.n += bits

    def bytes(self, total):
        pad = total * 8 - self.n
        assert pad >= 0, (self.n, total * 8)
        return ((self.v << pad)).to_bytes(total, "big")


def header(layer, version="mpeg1", bitrate_index=9, sr=0, padding=0, mode="stereo", mode_ext=0, crc=False):
    vbits = {"mpeg1": 3, "mpeg2": 2, "mpeg25": 0}[version]
    lbits = {1: 3, 2: 2, 3: 1}[layer]
    b1 = 0xE0 | (vbits << 3) | (lbits << 1) | (0 if crc else 1)
    b2 = (bitrate_index << 4) | (sr << 2) | (padding << 1)
    b3 = (MODES[mode] << 6) | (mode_ext << 4)
    return bytes([0xFF, b1, b2, b3])


def frame_bytes(layer, version, bitrate_index, sr, padding):
    hz = [44100, 48000, 32000][sr] >> (0 if version == "mpeg1" else 1) >> (1 if version == "mpeg25" else 0)
    kbps = 2 * HALFRATE[1 if version == "mpeg1" else 0][3 - layer][bitrate_index]
    samples = 384 if layer == 1 else 1152
    n = samples * kbps * 125 // hz
    if layer == 1:
        n &= ~3
    return n + (padding * (4 if layer == 1 else 1)), hz


def random_frame(rng, layer, version="mpeg1", bitrate_index=9, sr=0, mode="stereo", mode_ext=0, crc=False, padding=None, fill=None):
    padding = int(rng.integers(0, 2)) if padding is None else padding
    n, _ = frame_bytes(layer, version, bitrate_index, sr, padding)
    if fill is not None:
        body = bytes([fill]) * (n - 4)
    else:
        # bits are 1 with a probability drawn per frame: uniform bits make the allocation ask for more sample bits than the
        # frame has (the decoder then drops the frame, minimp3.d:1573), sparse ones give frames that fit -- both occur
        density = float(rng.choice([0.12, 0.2, 0.3, 0.4, 0.5]))
        body = np.packbits(rng.random((n - 4) * 8) < density).tobytes()
    return header(layer, version, bitrate_index, sr, padding, mode, mode_ext, crc) + body


def random_file(rng, layer, n_frames, version="mpeg1", bitrate_index=9, sr=0, mode="stereo", mode_ext=None, crc=False, vary_bitrate=False):
    out = []
    for _ in range(n_frames):
        bi = int(rng.integers(1, 15)) if vary_bitrate else bitrate_index
        me = int(rng.integers(0, 4)) if mode_ext is None else mode_ext
        out.append(random_frame(rng, layer, version, bi, sr, mode, me, crc))
    return b"".join(out)


def layer1_frame(rng, version="mpeg1", bitrate_index=12, sr=0, mode="stereo", mode_ext=0, max_bits=9, quiet=True):
    """One Layer I frame and what it means: returns (bytes, expected [channels][32 subbands][12 slots] float64).
    Joint stereo: subbands from the bound (4, 8, 12 or 16) up share allocation and samples, each channel keeps its own
    scalefactor."""
    nch = 1 if mode == "mono" else 2
    bound = 32 if mode != "joint" else 4 + 4 * mode_ext
    if nch == 1:
        bound = 0
    n, _ = frame_bytes(1, version, bitrate_index, sr, 0)
    budget = (n - 4) * 8
    b = MsbBits()
    # allocation: index a (0 = no samples, else a + 1 bits per sample); kept sparse enough for the frame size
    alloc = np.zeros((32, 2), np.int64)
    for sb in range(32):
        for ch in range(nch):
            if sb >= bound and ch == 1 and nch == 2:
                alloc[sb, 1] = alloc[sb, 0]
                continue
            alloc[sb, ch] = int(rng.integers(1, max_bits)) if rng.random() < (0.5 if sb < 12 else 0.15) else 0
    # trim to the budget
    def cost():
        bits = 0
        for sb in range(32):
            for ch in range(nch):
                shared = nch == 2 and sb >= bound
                if shared and ch == 1:
                    bits += 6 if alloc[sb, 0] else 0          # the second channel's scalefactor only
                    continue
                bits += 4
                if alloc[sb, ch]:
                    bits += 6 + 12 * (alloc[sb, ch] + 1)
        return bits
    sb = 31
    while cost() > budget:
        alloc[sb, :] = 0
        sb -= 1
    for sb in range(32):
        for ch in range(nch):
            if nch == 2 and sb >= bound and ch == 1:
                continue
            b.put(alloc[sb, ch], 4)
    scf = np.zeros((32, 2), np.int64)
    for sb in range(32):
        for ch in range(nch):
            if alloc[sb, ch]:
                scf[sb, ch] = int(rng.integers(20 if quiet else 0, 63))
                b.put(scf[sb, ch], 6)
    want = np.zeros((nch, 32, 12))
    for slot in range(12):
        for sb in range(32):
            for ch in range(nch):
                a = int(alloc[sb, ch])
                if not a:
                    continue
                nb = a + 1
                if nch == 2 and sb >= bound and ch == 1:
                    code = shared_code
                else:
                    code = int(rng.integers(0, (1 << nb) - 1))          # the all-ones code is forbidden
                    b.put(code, nb)
                    shared_code = code
                want[ch, sb, slot] = (code - ((1 << (nb - 1)) - 1)) * 2.0 / ((1 << nb) - 1) * 2.0 ** (-scf[sb, ch] / 3.0)
    assert b.n <= budget
    return header(1, version, bitrate_index, sr, 0, mode, mode_ext) + b.bytes(n - 4), want
