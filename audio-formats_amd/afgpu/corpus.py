"""Device-resident workloads of BASELINE.json: C2 / C3 / C4 and the C5 mixed corpus.

A *workload* is a set of transform-stage batches that sit in HBM (what the host front-ends would have
uploaded) plus the launches that decode them.  bench.py times `Workload.step()`; the tests run the small
numpy twins of the same generators against the oracle.

C5 (BASELINE configs[4]): 65 536 files, 40 % MP3 / 25 % Ogg Vorbis / 25 % FLAC / 10 % Opus (CELT-only, 960-sample
stereo frames, post-filter on 30 % of the frames, T in [15, 1022]); durations log-uniform in 4..30 s, seeds fixed.
Files are independent, so the corpus shards by file (sharding.lpt_partition on frames x channels) and a rank
walks its shard in waves of at most 8192 files so that a wave's planes fit one MI355X.
"""
import numpy as np

from . import (CELT_FRAME_DTYPE, FLAC_FRAME_DTYPE, FLAC_INDEPENDENT, FLAC_LEFT_SIDE, FLAC_MID_SIDE,
               FLAC_SUBFRAME_DTYPE, NUMERIC_TOLERANCE, VORBIS_LONG, VORBIS_NZ_EIGHTHS, Mp3Plan, VorbisPlan, celt_transform, flac_transform, flac_variants,
               get_numeric_mode, sharding, synthetic)

KIND_MP3, KIND_VORBIS, KIND_FLAC, KIND_CELT = 0, 1, 2, 3
KIND_NAMES = ["mp3", "vorbis", "flac", "celt"]

# algorithmic bytes per unit
MP3_BYTES_PER_GRCH = 2304 + 4 + 2304            # f32 spectrum + flag word in, f32 PCM out
FLAC_BYTES_PER_FRAME_REC = 32 + 2 * 68          # one frame record + two subframe records (stereo)
CELT_BYTES_PER_REC = 48

C5_FILES = 65536
# files resident at once on one GPU: a third of the corpus is ~200 GB of planes (the Opus members' serial per-stream chains
# take as long for 800 streams as for 2500, so fewer, larger waves pay: 8192 -> 24576 files took C5 from 304 to 200 ms)
C5_WAVE_FILES = 24576
C5_SEED = 0xC5
# Device time per decoded sample, ns, by kind (MP3, Vorbis, FLAC, CELT): each codec's kernel alone over its C5 members on one
# MI355X (profiles/r03_c5.json; round 2 sharded on samples, when a CELT sample cost 8 x an MP3 sample).  Only the ratios matter.
C5_COST_NS_PER_SAMPLE = np.array([1.85e-3, 1.87e-3, 1.41e-3, 2.75e-3])
C5_SPREAD_OPUS = 64             # the longest Opus files are dealt out evenly before the greedy pass (sharding.lpt_partition)


# --------------------------------------------------------------------------- manifest

def c5_manifest(n_files=C5_FILES, seed=C5_SEED):
    """Deterministic description of the mixed corpus: kind[i], units[i] (MP3 granules / Vorbis packets / FLAC frames /
    CELT frames of file i, all stereo) and work[i] = decoded samples (frames x channels)."""
    rng = np.random.default_rng(seed)
    kind = rng.choice(4, size=n_files, p=[0.40, 0.25, 0.25, 0.10]).astype(np.int8)
    seconds = np.exp(rng.uniform(np.log(4.0), np.log(30.0), n_files))
    units = np.empty(n_files, np.int64)
    m = kind == KIND_MP3
    units[m] = 2 * np.maximum(1, np.round(seconds[m] * 44100 / 1152)).astype(np.int64)          # granules (2 per frame)
    m = kind == KIND_VORBIS
    units[m] = 1 + np.maximum(1, np.round(seconds[m] * 44100 / 1024)).astype(np.int64)          # packets (first one primes)
    m = kind == KIND_FLAC
    units[m] = np.maximum(1, np.round(seconds[m] * 44100 / 4096)).astype(np.int64)              # frames of 4096
    m = kind == KIND_CELT
    units[m] = np.maximum(1, np.round(seconds[m] * 50)).astype(np.int64)                        # 20 ms frames at 48 kHz
    per_unit = np.array([576, 1024, 4096, 960], np.int64)[kind]
    work = units * per_unit * 2
    work[kind == KIND_VORBIS] -= 1024 * 2                                                        # the first packet delivers nothing
    return with_cost({"kind": kind, "seconds": seconds, "units": units, "work": work})


def with_cost(manifest):
    """Adds what the partition balances: cost[i] = predicted device time of file i (ns) and spread = the C5_SPREAD_OPUS
    longest Opus files (by frames), which no rank may collect."""
    kind, work, units = manifest["kind"], manifest["work"], manifest["units"]
    manifest["cost"] = work * C5_COST_NS_PER_SAMPLE[kind]
    opus = np.flatnonzero(kind == KIND_CELT)
    manifest["spread"] = opus[np.lexsort((opus, -units[opus]))][:C5_SPREAD_OPUS]
    return manifest


def c5_partition(manifest, world):
    """rank_of[file]: LPT on predicted device time, the longest Opus files spread first (a manifest without `cost`:
    on samples)."""
    return sharding.lpt_partition(manifest.get("cost", manifest["work"]), world, manifest.get("spread"))


def c5_imbalance(manifest, world):
    """max / mean predicted device time over the ranks"""
    return sharding.imbalance(manifest.get("cost", manifest["work"]), world, manifest.get("spread"))


def c5_shard_waves(manifest, rank, world, wave_files=C5_WAVE_FILES):
    """File indices of `rank`'s shard, cut into waves of at most wave_files files (ascending file order).  Every rank
    gets the same number of waves (the partition is deterministic, so each rank derives it from all shards' sizes):
    the ranks meet in a barrier around every wave's timed region."""
    rank_of = c5_partition(manifest, world)
    counts = np.bincount(rank_of, minlength=world)
    n_waves = max(1, int(-(-counts.max() // wave_files)))
    mine = np.flatnonzero(rank_of == rank)
    return [w for w in np.array_split(mine, n_waves)]


# --------------------------------------------------------------------------- numpy record generators
# Everything random about a file is drawn from a generator keyed by (seed, file id, field): a file's records and
# inputs are the same whichever rank, wave or position in a plane it lands on ("output identical for
# 1/2/4/8 GPUs").

_CELT_TAPS = np.array([[0.3066406250, 0.2170410156, 0.1296386719], [0.4638671875, 0.2680664062, 0.0],
                       [0.7998046875, 0.1000976562, 0.0]], np.float32)            # dopus.d:3382-3386


def _ids(n, file_ids):
    return np.arange(n) if file_ids is None else np.asarray(file_ids)


def celt_records(seed, frames_per_stream, file_ids=None, channels=2, p_transient=0.15, p_postfilter=0.3):
    """960-sample stereo CELT frames: transients (8 short blocks) on 15 % of the frames, a new post-filter on 30 %
    (T in [15, 1022], gain and tapset as parse_postfilter decodes them, dopus.d:3380-3418), period kept across frames
    without one (:3391).  Sequence (stream, channel) owns a contiguous run of records; output interleaved per stream.
    Returns (rec_base, recs, out_total, coef_floats); coefficients are generated separately."""
    frames_per_stream = np.asarray(frames_per_stream, np.int64)
    ns = len(frames_per_stream)
    ids = _ids(ns, file_ids)
    C = channels
    nf_total = int(frames_per_stream.sum())
    trans = np.empty(nf_total, bool)
    haspf = np.empty(nf_total, bool)
    period = np.empty(nf_total, np.int64)
    gain = np.empty(nf_total, np.float32)
    tapset = np.empty(nf_total, np.int64)
    at = 0
    for s in range(ns):
        nf = int(frames_per_stream[s])
        rng = np.random.default_rng([seed, int(ids[s]), 4])
        trans[at:at + nf] = rng.random(nf) < p_transient
        haspf[at:at + nf] = rng.random(nf) < p_postfilter
        period[at:at + nf] = rng.integers(15, 1023, nf)
        gain[at:at + nf] = (0.09375 * (rng.integers(0, 8, nf) + 1)).astype(np.float32)     # dopus.d:3400
        tapset[at:at + nf] = rng.integers(0, 3, nf)
        at += nf
    first = np.concatenate([[0], np.cumsum(frames_per_stream)[:-1]]).astype(np.int64)    # first frame of each stream
    stream_of = np.repeat(np.arange(ns), frames_per_stream)
    idx = np.arange(nf_total)
    last_pf = np.maximum.accumulate(np.where(haspf, idx, -1)) if nf_total else idx
    valid = last_pf >= first[stream_of]
    per = np.where(valid, period[np.maximum(last_pf, 0)], 0).astype(np.int32)
    g = np.where(haspf[:, None], gain[:, None] * _CELT_TAPS[tapset], 0).astype(np.float32)
    blocks = np.where(trans, 8, 1).astype(np.uint8)
    frame_in_stream = idx - first[stream_of]
    out_base = np.concatenate([[0], np.cumsum(frames_per_stream * 960 * C)[:-1]]).astype(np.int64)
    rec_first = np.concatenate([[0], np.cumsum(frames_per_stream * C)[:-1]]).astype(np.int64)
    recs = np.zeros(nf_total * C, CELT_FRAME_DTYPE)
    for c in range(C):
        pos = rec_first[stream_of] + c * frames_per_stream[stream_of] + frame_in_stream
        recs["coef_off"][pos] = pos.astype(np.uint64) * np.uint64(960)
        recs["out_off"][pos] = (out_base[stream_of] + frame_in_stream * 960 * C + c).astype(np.uint64)
        recs["out_stride"][pos] = C
        recs["frame_size"][pos] = 960
        recs["blocks"][pos] = blocks
        recs["pf_period_new"][pos] = per
        recs["pf_gains_new"][pos] = g
        recs["imdct_scale"][pos] = 1.0
    rb = np.empty(ns * C + 1, np.uint64)
    for c in range(C):
        rb[c:-1:C] = (rec_first + c * frames_per_stream).astype(np.uint64)
    rb[-1] = nf_total * C
    return rb, recs, int(nf_total * 960 * C), int(nf_total * C * 960)


def celt_coefs_numpy(seed, frames_per_stream, file_ids=None):
    k = np.arange(960, dtype=np.float64)
    tilt = 2000.0 * 10.0 ** (-(k / 960) * 2.0)
    ids = _ids(len(frames_per_stream), file_ids)
    out = [(np.random.default_rng([seed, int(ids[s]), 5]).standard_normal((2 * int(nf), 960)) * tilt).astype(np.float32).reshape(-1)
           for s, nf in enumerate(frames_per_stream)]
    return np.concatenate(out) if out else np.zeros(0, np.float32)


_LPC_POOL = {}


def _lpc_pool(seed):
    if seed not in _LPC_POOL:
        # quantised LPC of a STABLE AR(order) process -- with the residual clamp below every decoded sample stays
        # inside 16 bits per subframe (synthetic.stable_quantised_lpc: sum |h| <= 24, |residual| <= 1023)
        _LPC_POOL[seed] = {o: ([synthetic.stable_quantised_lpc(np.random.default_rng([seed, o, i]), o) for i in range(64)]) for o in (8, 12)}
        assert synthetic.FLAC_C4_L1_MAX * (synthetic.FLAC_C4_RESIDUAL_CLAMP + 1) < 32768
    return _LPC_POOL[seed]


def flac_records(seed, frames_per_file, file_ids=None, block_size=4096, bps=16):
    """Stereo FLAC transform-stage records for files of frames_per_file[i] frames: LPC order 8 (even file ids) / 12 (odd),
    60 % MID_SIDE / 25 % LEFT_SIDE / 15 % independent, side subframes flagged use64.
    Returns (frames, subframes); residuals are generated separately."""
    frames_per_file = np.asarray(frames_per_file, np.int64)
    ids = _ids(len(frames_per_file), file_ids)
    n_frames = int(frames_per_file.sum())
    pool = _lpc_pool(0xF1AC)
    tabs = {o: (np.stack([pool[o][i][0] for i in range(64)]), np.array([pool[o][i][1] for i in range(64)], np.uint8)) for o in (8, 12)}
    frames = np.zeros(n_frames, FLAC_FRAME_DTYPE)
    idx = np.arange(n_frames, dtype=np.uint64)
    frames["in_off"] = idx * np.uint64(block_size * 2)
    frames["out_off"] = idx * np.uint64(block_size * 2)
    frames["block_size"] = block_size
    frames["sf_index"] = (idx * 2).astype(np.uint32)
    frames["channels"] = 2
    frames["bps"] = bps
    subframes = np.zeros(n_frames * 2, FLAC_SUBFRAME_DTYPE)
    asg = np.empty(n_frames, np.int64)
    at = 0
    for k, nf in enumerate(frames_per_file):
        nf = int(nf)
        rng = np.random.default_rng([seed, int(ids[k]), 3])
        asg[at:at + nf] = rng.choice([FLAC_MID_SIDE, FLAC_LEFT_SIDE, FLAC_INDEPENDENT], size=nf, p=[0.6, 0.25, 0.15])
        order = 8 if int(ids[k]) % 2 == 0 else 12
        coefs, shifts = tabs[order]
        pick = rng.integers(0, 64, 2 * nf)
        sl = slice(2 * at, 2 * (at + nf))
        subframes["coef"][sl, :order] = coefs[pick]
        subframes["order"][sl] = order
        subframes["shift"][sl] = shifts[pick]
        at += nf
    frames["assignment"] = asg.astype(np.uint8)
    side = np.zeros(n_frames * 2, bool)
    side[1::2] = np.isin(asg, [FLAC_MID_SIDE, FLAC_LEFT_SIDE])
    subframes["use64"] = side.astype(np.uint8)
    return frames, subframes


def flac_residuals_numpy(seed, frames_per_file, file_ids=None, block_size=4096):
    ids = _ids(len(frames_per_file), file_ids)
    lim = synthetic.FLAC_C4_RESIDUAL_CLAMP
    out = [np.clip(np.rint(np.random.default_rng([seed, int(ids[k]), 6]).laplace(0.0, 32.0, int(nf) * 2 * block_size)), -lim, lim).astype(np.int32)
           for k, nf in enumerate(frames_per_file)]
    return np.concatenate(out) if out else np.zeros(0, np.int32)


def mp3_flag_plane(seed, granules_per_file, file_ids=None, p_event=0.04):
    """Flag words of stereo files (block-type sequences per channel, AFG_MP3_NZ_BANDS declared as the host parser does)."""
    ids = _ids(len(granules_per_file), file_ids)
    out = []
    for k, ng in enumerate(granules_per_file):
        rng = np.random.default_rng([seed, int(ids[k]), 0])
        f = np.zeros((int(ng), 2), np.uint32)
        for ch in range(2):
            bt, mixed = synthetic.mp3_block_types(rng, int(ng), p_event, 0.0)
            f[:, ch] = synthetic.mp3_flag_words(bt, mixed, synthetic.MP3_NZ_BANDS)
        out.append(f.reshape(-1))
    return np.concatenate(out) if out else np.zeros(0, np.uint32)


def mp3_coefs_numpy(seed, granules_per_file, file_ids=None):
    ids = _ids(len(granules_per_file), file_ids)
    tilt = synthetic.mp3_tilt()
    out = []
    for k, ng in enumerate(granules_per_file):
        c = np.random.default_rng([seed, int(ids[k]), 1]).standard_normal((int(ng) * 2, 576)).astype(np.float32) * tilt
        c[:, synthetic.MP3_CUTOFF_LINE:] = 0.0
        out.append(c.reshape(-1))
    return np.concatenate(out) if out else np.zeros(0, np.float32)


def vorbis_flag_plane(seed, packets_per_file, file_ids=None, p_short_run=0.02):
    """Packet flags of the files: prefixes of 16 long legal sequences (one family per file id mod 16)."""
    ids = _ids(len(packets_per_file), file_ids)
    longest = int(max(packets_per_file)) if len(packets_per_file) else 0
    if not longest:
        return np.zeros(0, np.uint8)
    cap = 1 << int(np.ceil(np.log2(max(longest, 16))))          # a family's sequence does not depend on who else is in the plane
    fams = {}
    out = []
    for k, n in enumerate(packets_per_file):
        f = int(ids[k]) % 16
        if f not in fams or len(fams[f]) < int(n):
            fams[f] = synthetic.vorbis_packet_flags(np.random.default_rng([seed, f, 2]), 2 * max(cap, 1 << 12), p_short_run)
        out.append(fams[f][:int(n)])
    return np.concatenate(out)


def vorbis_spec_numpy(seed, pflags, packets_per_file, file_ids=None, bs0=256, bs1=2048):
    ids = _ids(len(packets_per_file), file_ids)
    curve = {bs1 // 2: synthetic.vorbis_floor_curve(bs1 // 2), bs0 // 2: np.float32(0.25 * synthetic.VORBIS_LEVEL) * np.ones(bs0 // 2, np.float32)}
    out, pk = [], 0
    for k, n in enumerate(packets_per_file):
        rng = np.random.default_rng([seed, int(ids[k]), 7])
        for q in range(int(n)):
            n2 = (bs1 if pflags[pk + q] & VORBIS_LONG else bs0) // 2
            out.append((rng.standard_normal((2, n2)).astype(np.float32) * curve[n2]).reshape(-1))
        pk += int(n)
    return np.concatenate(out) if out else np.zeros(0, np.float32)


# --------------------------------------------------------------------------- device-resident parts

class Part:
    """One codec's share of a workload: planes in HBM + the launch.  `host=True` builds the inputs with the per-file
    numpy generators (identical for a file wherever it lands; small batches, tests); otherwise the big planes are drawn
    on the device by torch (bench sizes) and only the small records come from numpy."""
    name = ""
    kernel = ""
    samples = 0                 # decoded samples per launch
    alg_bytes = 0               # algorithmic bytes per launch
    survey_bytes = None         # the same at SURVEY 8(d)'s per-unit figure, where the launch's input format moves fewer
    units = 0

    def launch(self, stream):   # pragma: no cover - interface
        raise NotImplementedError

    def check(self, checker, n_files=1):  # pragma: no cover - interface
        raise NotImplementedError

    def file_bounds(self):      # pragma: no cover - interface
        """[n_files + 1] offsets of every file's output in the part's output plane."""
        raise NotImplementedError

    def file_outputs(self):
        """Per-file numpy copies of the output plane (tests)."""
        b = self.file_bounds()
        out = self.out_plane().cpu().numpy()
        return [out[int(b[i]):int(b[i + 1])] for i in range(len(b) - 1)]


class Mp3Part(Part):
    name = "mp3"
    kernel = "mp3_tolerance_kernel (tolerance mode: csrc/mp3_kernel.h with fused multiply-adds; exact mode: mp3_transform_kernel)"

    def __init__(self, seed, granules_per_file, device, seg=0, file_ids=None, host=False):
        import torch
        g = np.asarray(granules_per_file, np.uint32)
        self.granules = g
        self.channels = np.full(len(g), 2, np.uint8)
        self.plan = Mp3Plan(self.granules, self.channels, seg)
        blocks = self.plan.blocks
        if host:
            self.coef = torch.from_numpy(mp3_coefs_numpy(seed, g, file_ids)).to(device)
        else:
            self.coef = torch.empty(blocks * 576, dtype=torch.float32, device=device)
            tilt = torch.from_numpy(synthetic.mp3_tilt()).to(device)
            gen = torch.Generator(device=device)
            gen.manual_seed(int(seed))
            rows = self.coef.view(-1, 576)
            step = 1 << 19
            for r0 in range(0, rows.shape[0], step):
                v = rows[r0:r0 + step]
                v.normal_(generator=gen)
                v.mul_(tilt)
                v[:, synthetic.MP3_CUTOFF_LINE:] = 0.0            # +0.0 (a negative draw times 0 is -0.0)
        flags_np = mp3_flag_plane(seed, g, file_ids)
        self.flags = torch.from_numpy(flags_np.view(np.int32)).to(device)
        # the same flags without the AFG_MP3_NZ_BANDS declaration: the kernel then fetches all 32 subbands
        self.flags_full = torch.from_numpy((flags_np & np.uint32(0x00ffffff)).view(np.int32)).to(device)
        self.pcm = torch.empty_like(self.coef)
        self.units = int(blocks)
        self.samples = int(blocks) * 576
        self.alg_bytes = int(blocks) * MP3_BYTES_PER_GRCH

    def launch(self, stream, full_fetch=False):
        self.plan.transform(self.coef, self.flags_full if full_fetch else self.flags, self.pcm, None, stream)

    def out_plane(self):
        return self.pcm

    def file_bounds(self):
        return np.concatenate([[0], np.cumsum(self.granules.astype(np.int64) * 2 * 576)])

    def check(self, checker, n_files=1):
        nb = int(self.granules[:n_files].astype(np.int64).sum()) * 2
        want = checker.mp3_transform(self.granules[:n_files], self.channels[:n_files], self.coef[:nb * 576].cpu().numpy(),
                                       self.flags[:nb].cpu().numpy().view(np.uint32))
        got = self.pcm[:nb * 576].cpu().numpy()
        return _float_parity(got, want, get_numeric_mode() == NUMERIC_TOLERANCE)

    def check_file(self, checker, f):
        """file f on its own (any position in the plane: the last files of a full-size batch sit beyond 2^32 bytes)"""
        b0 = int(self.granules[:f].astype(np.int64).sum()) * 2
        nb = int(self.granules[f]) * 2
        want = checker.mp3_transform(self.granules[f:f + 1], self.channels[f:f + 1], self.coef[b0 * 576:(b0 + nb) * 576].cpu().numpy(),
                                     self.flags[b0:b0 + nb].cpu().numpy().view(np.uint32))
        return _float_parity(self.pcm[b0 * 576:(b0 + nb) * 576].cpu().numpy(), want, get_numeric_mode() == NUMERIC_TOLERANCE)


class VorbisPart(Part):
    name = "vorbis"
    kernel = ("vorbis_walk_kernel (tolerance mode: IMDCT-2048 as one radix 8 x 8 x 8 FFT, csrc/vorbis_walk.hip; exact mode: vorbis_wave_kernel, "
              "the reference's 8-step algorithm)")

    def __init__(self, seed, packets_per_file, device, seg=0, bs0=256, bs1=2048, file_ids=None, host=False, declare_zero_tail=True):
        import torch
        n = np.asarray(packets_per_file, np.uint32)
        pflags = vorbis_flag_plane(seed, n, file_ids)
        # what the host parser declares from the residue's end (host/afg_vorbis_front.cpp): the floor curve is zero from bin
        # 743 of 1024 up, i.e. the last two eighths of a long block's spectrum
        nz_bins = int(np.flatnonzero(synthetic.vorbis_floor_curve(bs1 // 2))[-1]) + 1
        import os
        if os.environ.get("AFG_VORBIS_DECLARE", "1") == "0":               # A/B only: the same spectra, nothing declared
            declare_zero_tail = False
        self.nz_eighths = -(-nz_bins * 8 // (bs1 // 2)) if declare_zero_tail else 8
        if self.nz_eighths < 8:
            pflags = np.where(pflags & VORBIS_LONG, pflags | np.uint8(VORBIS_NZ_EIGHTHS(self.nz_eighths)), pflags).astype(np.uint8)
        nf = len(n)
        self.plan = VorbisPlan(n, np.full(nf, 2, np.uint8), np.full(nf, bs0, np.uint16), np.full(nf, bs1, np.uint16), pflags, seg)
        # the same packets with nothing declared (every long block's whole spectrum fetched): built on first use (launch(full_fetch=True))
        self._plan_full, self._plan_full_args = None, (n, np.full(nf, 2, np.uint8), np.full(nf, bs0, np.uint16), np.full(nf, bs1, np.uint16),
                                                       (pflags & np.uint8(0x0f)).astype(np.uint8), seg)       # bits 4-7: AFG_VORBIS_NZ_EIGHTHS
        if host:
            self.spec = torch.from_numpy(vorbis_spec_numpy(seed, pflags, n, file_ids, bs0, bs1)).to(device)
            assert self.spec.numel() == self.plan.spec_floats
        else:
            self.spec = torch.empty(self.plan.spec_floats, dtype=torch.float32, device=device)
            gen = torch.Generator(device=device)
            gen.manual_seed(int(seed))
            chunk = 1 << 28
            for o in range(0, self.plan.spec_floats, chunk):
                self.spec[o:o + chunk].normal_(generator=gen)
            # the same shaping as vorbis_spec_numpy: the floor curve (zeros above 16 kHz) on long blocks, 0.25 on short ones; a
            # plane of 64-float rows, each row knowing which 64 bins of its channel it holds
            rows_long, rows_short, row = bs1 // 128, bs0 // 128, 64
            is_long = (pflags & VORBIS_LONG) != 0
            per_packet = np.where(is_long, 2 * rows_long, 2 * rows_short).astype(np.int64)
            first = np.concatenate([[0], np.cumsum(per_packet)])
            kind = np.full(int(first[-1]), rows_long, np.int16)                       # rows_long = "a short block's row"
            long_first = first[:-1][is_long]
            if len(long_first):
                idx = (long_first[:, None] + np.arange(2 * rows_long)[None, :]).reshape(-1)
                kind[idx] = np.tile(np.arange(2 * rows_long) % rows_long, len(long_first))
            table = np.concatenate([synthetic.vorbis_floor_curve(bs1 // 2).reshape(rows_long, row), np.full((1, row), 0.25 * synthetic.VORBIS_LEVEL, np.float32)])
            d_table = torch.from_numpy(table).to(device)
            d_kind = torch.from_numpy(kind).to(device)
            view = self.spec.view(-1, row)
            assert view.shape[0] == len(kind)
            step = 1 << 22
            for o in range(0, len(kind), step):
                view[o:o + step].mul_(d_table[d_kind[o:o + step].long()])
            del d_kind
        self.out = torch.empty(self.plan.out_floats, dtype=torch.float32, device=device)
        self.units = int(self.plan.total_packets) * 2
        self.samples = int(self.plan.out_floats)
        # declared-empty eighths of long blocks are not read: they are not algorithmic bytes of this launch (survey_bytes keeps
        # SURVEY 8(d)'s per-packet figure, which counts the whole spectrum)
        self.survey_bytes = 4 * int(self.plan.spec_floats) + int(self.plan.total_packets) + 4 * int(self.plan.out_floats)
        n_long = int(((pflags & VORBIS_LONG) != 0).sum())
        self.alg_bytes = self.survey_bytes - 4 * n_long * 2 * (bs1 // 2) * (8 - self.nz_eighths) // 8

    def launch(self, stream, full_fetch=False):
        if full_fetch:
            if self._plan_full is None:
                self._plan_full = VorbisPlan(*self._plan_full_args)
            self._plan_full.transform(self.spec, self.out, stream)
        else:
            self.plan.transform(self.spec, self.out, stream)

    def out_plane(self):
        return self.out

    def file_bounds(self):
        _, oo = self.plan.offsets()
        first = np.concatenate([[0], np.cumsum(self.plan.packets.astype(np.int64))])
        return np.array([int(oo[k]) if k < self.plan.total_packets else self.plan.out_floats for k in first], np.int64)

    def check(self, checker, n_files=1):
        p = self.plan
        so, oo = p.offsets()
        npk = int(p.packets[:n_files].astype(np.int64).sum())
        s_end = int(so[npk]) if p.total_packets > npk else p.spec_floats
        o_end = int(oo[npk]) if p.total_packets > npk else p.out_floats
        want = checker.vorbis_transform(p.packets[:n_files], p.channels[:n_files], p.bs0[:n_files], p.bs1[:n_files],
                                          p.pflags[:npk], so[:npk], oo[:npk], self.spec[:s_end].cpu().numpy(), o_end)
        return _float_parity(self.out[:o_end].cpu().numpy(), want, get_numeric_mode() == NUMERIC_TOLERANCE)

    def check_file(self, checker, f):
        p = self.plan
        so, oo = p.offsets()
        k0 = int(p.packets[:f].astype(np.int64).sum())
        k1 = k0 + int(p.packets[f])
        s0, o0 = int(so[k0]), int(oo[k0])
        s1 = int(so[k1]) if p.total_packets > k1 else p.spec_floats
        o1 = int(oo[k1]) if p.total_packets > k1 else p.out_floats
        want = checker.vorbis_transform(p.packets[f:f + 1], p.channels[f:f + 1], p.bs0[f:f + 1], p.bs1[f:f + 1], p.pflags[k0:k1],
                                        so[k0:k1] - np.uint64(s0), oo[k0:k1] - np.uint64(o0), self.spec[s0:s1].cpu().numpy(), o1 - o0)
        return _float_parity(self.out[o0:o1].cpu().numpy(), want, get_numeric_mode() == NUMERIC_TOLERANCE)


class FlacPart(Part):
    name, kernel = "flac", "flac_restore1_kernel"

    def __init__(self, seed, frames_per_file, device, block_size=4096, file_ids=None, host=False, res16=None):
        import os
        import torch
        fpf = np.asarray(frames_per_file, np.int64)
        self.frames_per_file = fpf
        self.frames_np, self.sub_np = flac_records(seed, fpf, file_ids, block_size)
        self.n_frames = len(self.frames_np)
        total = self.n_frames * 2 * block_size
        self.block_size = block_size
        # residual rows as int16 (AFG_FLAC_ROW16): what a 16-bit file's residuals fit into; the same values as
        # the int32 planes of round 1 (AFG_FLAC_RES32=1 keeps those for an A/B)
        self.res16 = (not os.environ.get("AFG_FLAC_RES32")) if res16 is None else bool(res16)
        if self.res16:
            assert block_size % 8 == 0
            self.frames_np["res16"] = 1            # in_off (2 * block_size per frame) now counts int16 elements
        if host:
            r = flac_residuals_numpy(seed, fpf, file_ids, block_size)
            if self.res16:
                assert np.abs(r).max(initial=0) < 32768
                r = r.astype(np.int16).view(np.int32)
            self.res = torch.from_numpy(r).to(device)
        elif self.res16:
            gen = torch.Generator(device=device)
            gen.manual_seed(int(seed))
            r16 = torch.empty(total, dtype=torch.int16, device=device)
            chunk = 1 << 28
            for o in range(0, total, chunk):
                k = min(chunk, total - o)
                e = torch.empty(k, dtype=torch.float32, device=device).exponential_(1.0 / 32.0, generator=gen)
                sgn = torch.empty(k, dtype=torch.float32, device=device).uniform_(-1.0, 1.0, generator=gen).sign_()
                r16[o:o + k] = (e * sgn).round_().clamp_(-synthetic.FLAC_C4_RESIDUAL_CLAMP, synthetic.FLAC_C4_RESIDUAL_CLAMP).to(torch.int16)
                del e, sgn
            self.res = r16.view(torch.int32)
        else:
            gen = torch.Generator(device=device)
            gen.manual_seed(int(seed))
            self.res = torch.empty(total, dtype=torch.int32, device=device)
            chunk = 1 << 28
            for o in range(0, total, chunk):
                k = min(chunk, total - o)
                e = torch.empty(k, dtype=torch.float32, device=device).exponential_(1.0 / 32.0, generator=gen)
                sgn = torch.empty(k, dtype=torch.float32, device=device).uniform_(-1.0, 1.0, generator=gen).sign_()
                self.res[o:o + k] = (e * sgn).round_().clamp_(-synthetic.FLAC_C4_RESIDUAL_CLAMP, synthetic.FLAC_C4_RESIDUAL_CLAMP).to(torch.int32)
                del e, sgn
        self.d_frames = torch.from_numpy(self.frames_np.view(np.uint8).copy()).to(device)
        self.d_sub = torch.from_numpy(self.sub_np.view(np.uint8).copy()).to(device)
        # which instantiations of the restore kernel these records populate: known from the host copy of the records, as
        # a parser knows it (AFG_FLAC_ALL_VARIANTS=1: the plain entry, all 16 launched on one stream)
        self.variants = None if os.environ.get("AFG_FLAC_ALL_VARIANTS") else flac_variants(self.frames_np, self.sub_np)
        self.out = torch.empty(total, dtype=torch.int32, device=device)
        self.units = self.n_frames * 2
        self.samples = total
        # int32 residual + int32 output = 8 B / sample; int16 residual rows make it 6 B / sample, and that is
        # what the roofline is priced on (the 8 B figure is kept beside it, not used for `frac`)
        self.survey_bytes = 8 * total + self.n_frames * FLAC_BYTES_PER_FRAME_REC
        self.alg_bytes = (6 if self.res16 else 8) * total + self.n_frames * FLAC_BYTES_PER_FRAME_REC

    def launch(self, stream):
        flac_transform(self.n_frames, self.d_frames, self.d_sub, self.res, self.out, None, stream, variants=self.variants)

    def out_plane(self):
        return self.out

    def file_bounds(self):
        return np.concatenate([[0], np.cumsum(self.frames_per_file * 2 * self.block_size)])

    def check(self, checker, n_files=2):
        nchk = int(self.frames_per_file[:n_files].sum())
        cnt = nchk * 2 * self.block_size
        words = cnt // 2 if self.res16 else cnt
        want = checker.flac_transform(self.frames_np[:nchk], self.sub_np[:2 * nchk], self.res[:words].cpu().numpy(), cnt)
        got = self.out[:cnt].cpu().numpy()
        bad = int((got != want).sum())
        # a stable AR process -- every sample (left-justified 16-bit PCM) inside 17 bits, as in a real file
        peak = int(np.abs(want.astype(np.int64) >> (32 - int(self.frames_np["bps"][0]))).max(initial=0))
        assert peak < (1 << 16), f"C4 generator: decoded sample magnitude {peak} is outside 17 bits"
        return {"samples": int(cnt), "mismatches": bad, "rms_error": 0.0 if bad == 0 else None, "max_abs_error": 0.0 if bad == 0 else None,
                "peak_sample_magnitude": peak}

    def check_file(self, checker, f):
        fr0 = int(self.frames_per_file[:f].sum())
        nfr = int(self.frames_per_file[f])
        frames = self.frames_np[fr0:fr0 + nfr].copy()
        w0, o0 = int(frames["in_off"][0]), int(frames["out_off"][0])
        cnt = nfr * 2 * self.block_size
        frames["in_off"] -= np.uint64(w0)
        frames["out_off"] -= np.uint64(o0)
        frames["sf_index"] -= np.uint32(2 * fr0)
        res = (self.res[w0 // 2:(w0 + cnt) // 2] if self.res16 else self.res[w0:w0 + cnt]).cpu().numpy()   # in_off: int16 / int32 units
        want = checker.flac_transform(frames, self.sub_np[2 * fr0:2 * (fr0 + nfr)], res, cnt)
        got = self.out[o0:o0 + cnt].cpu().numpy()
        bad = int((got != want).sum())
        return {"samples": int(cnt), "mismatches": bad, "rms_error": 0.0 if bad == 0 else None, "max_abs_error": 0.0 if bad == 0 else None}


class CeltPart(Part):
    name = "celt"
    kernel = ("celt_walk_kernel (tolerance mode; exact mode: celt_imdct_kernel + celt_postfilter_kernel + celt_deemph_kernel, "
              "or celt_stream_kernel + celt_deemph_kernel)")
    overlap = True

    def __init__(self, seed, frames_per_file, device, file_ids=None, host=False):
        import torch
        fpf = np.asarray(frames_per_file, np.int64)
        self.frames_per_file = fpf
        self.rb_np, self.recs_np, out_total, coef_floats = celt_records(seed, fpf, file_ids)
        if host:
            self.coef = torch.from_numpy(celt_coefs_numpy(seed, fpf, file_ids)).to(device)
        else:
            self.coef = torch.empty(coef_floats, dtype=torch.float32, device=device)
            k = np.arange(960, dtype=np.float64)
            tilt = torch.from_numpy((2000.0 * 10.0 ** (-(k / 960) * 2.0)).astype(np.float32)).to(device)
            gen = torch.Generator(device=device)
            gen.manual_seed(int(seed))
            rows = self.coef.view(-1, 960)
            step = 1 << 19
            for r0 in range(0, rows.shape[0], step):
                v = rows[r0:r0 + step]
                v.normal_(generator=gen)
                v.mul_(tilt)
        self.d_rb = torch.from_numpy(self.rb_np.view(np.int64)).to(device)
        self.d_recs = torch.from_numpy(self.recs_np.view(np.uint8).copy()).to(device)
        self.out = torch.empty(out_total, dtype=torch.float32, device=device)
        self.n_chan = len(self.rb_np) - 1
        self.units = len(self.recs_np)
        self.samples = out_total
        self.alg_bytes = 8 * out_total + CELT_BYTES_PER_REC * len(self.recs_np)

    def launch(self, stream, tail=None):
        celt_transform(self.n_chan, self.d_rb, self.d_recs, self.coef, self.out, None, stream, tail)

    def out_plane(self):
        return self.out

    def file_bounds(self):
        return np.concatenate([[0], np.cumsum(self.frames_per_file * 960 * 2)])

    def check(self, checker, n_files=1):
        nrec = int(self.frames_per_file[:n_files].sum()) * 2
        rb = self.rb_np[:2 * n_files + 1].copy()
        rb[-1] = nrec
        want = checker.celt_transform(rb, self.recs_np[:nrec], self.coef[:nrec * 960].cpu().numpy(), nrec * 960)
        return _float_parity(self.out[:nrec * 960].cpu().numpy(), want, get_numeric_mode() == NUMERIC_TOLERANCE, checker)

    def check_file(self, checker, f):
        """stereo file f on its own: its two channel sequences, records re-based to the file's first coefficient / sample"""
        r0, r1 = int(self.rb_np[2 * f]), int(self.rb_np[2 * f + 2])
        recs = self.recs_np[r0:r1].copy()
        c0, o0 = int(recs["coef_off"].min()), int(recs["out_off"].min())
        recs["coef_off"] -= np.uint64(c0)
        recs["out_off"] -= np.uint64(o0)
        rb = (self.rb_np[2 * f:2 * f + 3] - np.uint64(r0)).astype(np.uint64)
        n = (r1 - r0) * 960
        want = checker.celt_transform(rb, recs, self.coef[c0:c0 + n].cpu().numpy(), n)
        return _float_parity(self.out[o0:o0 + n].cpu().numpy(), want, get_numeric_mode() == NUMERIC_TOLERANCE, checker)


TOLERANCE_RMS = 1e-5            # north_star: float output within 1e-5 RMS of the reference decoders


def _float_parity(got, want, tolerance=False, checker=None):
    """Parity record of a float plane against the oracle.  `mismatches` is what callers gate on: the samples whose bits
    differ -- or, for a stage run in tolerance mode (afg.h AFG_NUMERIC_TOLERANCE: Opus/CELT), 0 when the RMS error is
    within TOLERANCE_RMS (else the bitwise count); such a record also carries the share of samples that land on a
    neighbouring int16 after OpusFile.readFrame's conversion."""
    diff = got.astype(np.float64) - want.astype(np.float64)
    bits = int((got.view(np.uint32) != want.view(np.uint32)).sum())
    rms = float(np.sqrt(np.mean(diff ** 2))) if got.size else 0.0
    rec = {"samples": int(got.size), "mismatches": bits, "rms_error": rms,
           "max_abs_error": float(np.abs(diff).max()) if got.size else 0.0}
    if tolerance:
        rec.update({"mode": "tolerance", "tolerance_rms": TOLERANCE_RMS, "bitwise_mismatches": bits,
                    "rms_signal": float(np.sqrt(np.mean(want.astype(np.float64) ** 2))) if got.size else 0.0,
                    "mismatches": 0 if (rms <= TOLERANCE_RMS and not np.isnan(got).any()) else max(bits, 1)})
        if checker is not None and got.size:
            gi, wi = checker.opus_output(got)[0], checker.opus_output(want)[0]
            step = np.abs(gi.astype(np.int32) - wi.astype(np.int32))
            rec.update({"int16_flip_rate": float((step != 0).mean()), "int16_max_step": int(step.max())})
    return rec


class Workload:
    """Parts that are resident together; a step launches each once, in order, on one stream."""

    def __init__(self, name, parts):
        self.name = name
        self.parts = parts

    @property
    def samples(self):
        return sum(p.samples for p in self.parts)

    @property
    def alg_bytes(self):
        return sum(p.alg_bytes for p in self.parts)

    def step_side_by_side(self, stream, lanes, side=None, events=None, order=None):
        """One launch of every part, part i on lanes[i % len(lanes)]: forked from `stream`, joined back into it.  The
        throughput kernels have different bottlenecks (MP3 issue, Vorbis / FLAC their access patterns); side by side they
        fill each other's gaps.  events as in step(): recorded on the stream each part runs on (overlapped spans)."""
        import torch
        fork = torch.cuda.Event()
        fork.record(stream)
        idx = list(range(len(self.parts))) if order is None else list(order)
        for n, i in enumerate(idx):
            p, s = self.parts[i], lanes[n % len(lanes)]
            s.wait_event(fork)
            if events is not None:
                events[i][0].record(s)
            tail = side is not None and getattr(p, "overlap", False)
            if tail:
                p.launch(s, side)
                j = torch.cuda.Event()
                j.record(side)
                s.wait_event(j)
            else:
                p.launch(s)
            if events is not None:
                events[i][1].record(s)
            done = torch.cuda.Event()
            done.record(s)
            stream.wait_event(done)

    def step(self, stream, events=None, side=None):
        """One launch of every part.  events: optional list of (start, end) torch events per part, recorded on the
        stream the part is launched on.  side: optional second stream; parts marked `overlap` (the CELT part of the
        mixed corpus: ~1600 long serial chains that occupy a fraction of the device for their whole length) are
        launched there, forked from and joined back into `stream`, and run beside the other codecs' kernels."""
        import os
        import torch
        # AFG_C5_ORDER (development, tools/gpu_c5_order.sh): "serial" keeps the Opus members on `stream`; a comma list of
        # part names fixes the launch order (default: the overlapped part first, then the parts as built)
        knob = os.environ.get("AFG_C5_ORDER", "")
        use_side = side is not None and any(getattr(p, "overlap", False) for p in self.parts) and knob != "serial"
        order = list(range(len(self.parts)))
        if use_side:
            # the part with the serial tail goes first: its record-parallel kernel has the device to itself for its ~2 ms on
            # `stream`, its per-sequence passes then run on `side` as the oldest wavefronts beside the throughput kernels
            # of the other parts (afg_celt_transform_streams_hip)
            order.sort(key=lambda i: not getattr(self.parts[i], "overlap", False))
        if "," in knob:
            names = knob.split(",")
            order.sort(key=lambda i: names.index(self.parts[i].name) if self.parts[i].name in names else len(names))
        for i in order:
            p = self.parts[i]
            tail = use_side and getattr(p, "overlap", False)
            if events is not None:
                events[i][0].record(stream)
            if tail:
                p.launch(stream, side)
            else:
                p.launch(stream)
            if events is not None:
                events[i][1].record(side if tail else stream)
        if use_side:
            join = torch.cuda.Event()
            join.record(side)
            stream.wait_event(join)


# BASELINE configs[1..3] at full size: 1024 x 60 s MP3, 1024 x 2584-packet Vorbis, 4096 x 323-frame FLAC
C2_FILES, C2_GRANULES = 1024, 2 * 2297
C3_FILES, C3_PACKETS = 1024, 2584
C4_FILES, C4_FRAMES = 4096, 323


def build_c234(device, rank=0, which=("mp3", "vorbis", "flac"), files=C2_FILES, seg=0):
    """The headline workload: C2 + C3 + C4 resident together (43 + 40 + 87 GB of planes at full size).  `files` scales
    all three (C4 has 4 x files)."""
    parts = []
    if "mp3" in which:
        parts.append(Mp3Part(0xA0D10 + 7919 * rank, np.full(files, C2_GRANULES), device, seg))
    if "vorbis" in which:
        parts.append(VorbisPart(0x0662 + 7919 * rank, np.full(files, C3_PACKETS), device))
    if "flac" in which:
        parts.append(FlacPart(0xF1AC + 7919 * rank, np.full(4 * files, C4_FRAMES), device))
    return Workload("+".join(which), parts)


def build_c5_wave(manifest, file_ids, device, seed=C5_SEED, host=False):
    """One wave of the mixed corpus: the files `file_ids` as up to four device-resident parts.  Every part keeps
    `file_ids` (the corpus ids of its files, in plane order).  host=True: inputs from the per-file numpy generators."""
    file_ids = np.asarray(file_ids)
    kind = manifest["kind"][file_ids]
    units = manifest["units"][file_ids]
    key = int(file_ids[0]) if len(file_ids) else 0
    parts = []
    for k, make in ((KIND_MP3, lambda u, ids: Mp3Part(seed if host else seed * 1000003 + key, u, device, 0, ids, host)),
                    (KIND_VORBIS, lambda u, ids: VorbisPart(seed if host else seed * 1000033 + key, u, device, 0, 256, 2048, ids, host)),
                    (KIND_FLAC, lambda u, ids: FlacPart(seed if host else seed * 1000037 + key, u, device, 4096, ids, host)),
                    (KIND_CELT, lambda u, ids: CeltPart(seed if host else seed * 1000039 + key, u, device, ids, host))):
        m = kind == k
        if m.any():
            part = make(units[m], file_ids[m])
            part.file_ids = file_ids[m]
            parts.append(part)
    return Workload("c5-wave", parts)
