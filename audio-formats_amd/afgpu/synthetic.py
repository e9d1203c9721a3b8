"""Synthetic, seeded transform-stage batches.

These stand in for what the host parsers produce right before the transform
stage: dequantised MP3 spectra + granule flags, Vorbis floor*residue spectra +
packet flags, FLAC residual planes + subframe records.  numpy versions feed the
parity tests; the torch versions build the BASELINE.json workloads directly in
HBM for bench.py.
"""
import numpy as np

from . import (CELT_FRAME_DTYPE, FLAC_FRAME_DTYPE, FLAC_INDEPENDENT, FLAC_LEFT_SIDE, FLAC_MID_SIDE, FLAC_RIGHT_SIDE,
               FLAC_SUBFRAME_DTYPE, VORBIS_LONG, VORBIS_NEXT, VORBIS_PREV, mp3_flags)

# ------------------------------------------------------------------ MP3 ------

MP3_CUTOFF_LINE = 418            # ~16 kHz at 44.1 kHz: lines above are zero at 128 kbps


# Level of the synthetic spectra (round 5): N(0, 1) lines under these shapes decoded to 8.2 (MP3) and 6.6 (Vorbis) times full
# scale; scaled so that the PCM has an encoder's level, rms about 0.05 -- the 1e-5 tolerance is then an absolute one on a
# signal inside full scale (timing does not depend on the values)
MP3_LEVEL = 0.05 / 8.18
VORBIS_LEVEL = 0.05 / 6.65


def mp3_tilt():
    k = np.arange(576, dtype=np.float64)
    t = MP3_LEVEL * 2.0 ** (-k / 48.0)
    t[MP3_CUTOFF_LINE:] = 0.0
    return t.astype(np.float32)


def mp3_block_types(rng, n, p_event=0.02, p_mixed=0.0):
    """Legal block-type sequence of one channel: 0* (1 2+ 3) 0* ...  Returns (block_type, mixed)."""
    bt = np.zeros(n, np.uint8)
    mixed = np.zeros(n, bool)
    starts = np.flatnonzero(rng.random(n) < p_event)
    pos = 0
    for s in starts:
        if s < pos:
            continue
        nshort = int(rng.integers(1, 4))
        if s + nshort + 2 > n:
            break
        bt[s] = 1
        bt[s + 1:s + 1 + nshort] = 2
        bt[s + 1 + nshort] = 3
        if p_mixed > 0:
            mixed[s + 1:s + 1 + nshort] = rng.random(nshort) < p_mixed
        pos = s + nshort + 2
    return bt, mixed


MP3_NZ_BANDS = (MP3_CUTOFF_LINE + 17) // 18          # subbands that can hold nonzero lines below the cut-off


def mp3_flag_words(bt, mixed, nz_bands=None):
    """AFG_MP3_FLAGS for MPEG-1 (n_long_bands = 2 for mixed blocks, minimp3.d:1218), optionally with
    AFG_MP3_NZ_BANDS(nz_bands): the spectra above that subband are +0.0 and need not be fetched."""
    n_long = np.where(mixed & (bt == 2), 2, 0).astype(np.uint32)
    aa = np.where(bt == 2, n_long.astype(np.int64) - 1, 31)
    w = (bt.astype(np.uint32) | (n_long << 8) | ((aa + 1).astype(np.uint32) << 16)).astype(np.uint32)
    if nz_bands is not None:
        w = w | np.uint32((int(nz_bands) + 1) << 24)
    return w


def mp3_batch(seed, granules, channels, p_event=0.05, p_mixed=0.3, amplitude=1.0, declare_nz=False):
    """numpy batch: returns (coef[blocks*576] f32, flags[blocks] u32).  The lines above the cut-off are +0.0;
    declare_nz puts AFG_MP3_NZ_BANDS into the flag words."""
    granules = np.asarray(granules, np.uint32)
    channels = np.asarray(channels, np.uint8)
    tilt = mp3_tilt()
    coefs, flags = [], []
    for s, (ng, nc) in enumerate(zip(granules, channels)):
        rng = np.random.default_rng([seed, s])
        c = rng.standard_normal((int(ng), int(nc), 576)).astype(np.float32) * tilt * np.float32(amplitude)
        c[..., MP3_CUTOFF_LINE:] = 0.0                        # +0.0, not the -0.0 a negative draw times 0 leaves
        f = np.zeros((int(ng), int(nc)), np.uint32)
        for ch in range(int(nc)):
            bt, mixed = mp3_block_types(rng, int(ng), p_event, p_mixed)
            f[:, ch] = mp3_flag_words(bt, mixed, MP3_NZ_BANDS if declare_nz else None)
        coefs.append(c.reshape(-1))
        flags.append(f.reshape(-1))
    if not coefs:
        return np.zeros(0, np.float32), np.zeros(0, np.uint32)
    return np.concatenate(coefs), np.concatenate(flags)


def mp3_batch_device(seed, n_files, granules_per_file, device, p_event=0.04, files_per_chunk=32):
    """C2-shaped stereo batch built in HBM with torch.  Returns (coef, flags) CUDA tensors."""
    import torch
    blocks = n_files * granules_per_file * 2
    coef = torch.empty(blocks * 576, dtype=torch.float32, device=device)
    tilt = torch.from_numpy(mp3_tilt()).to(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    per_file = granules_per_file * 2 * 576
    for f0 in range(0, n_files, files_per_chunk):
        f1 = min(n_files, f0 + files_per_chunk)
        view = coef[f0 * per_file:f1 * per_file].view(-1, 576)
        view.normal_(generator=gen)
        view.mul_(tilt)
        view[:, MP3_CUTOFF_LINE:] = 0.0                       # +0.0 (a negative draw times 0 is -0.0)
    flags_np = np.zeros((n_files, granules_per_file, 2), np.uint32)
    for f in range(n_files):
        rng = np.random.default_rng([seed, f])
        for ch in range(2):
            bt, mixed = mp3_block_types(rng, granules_per_file, p_event, 0.0)
            flags_np[f, :, ch] = mp3_flag_words(bt, mixed, MP3_NZ_BANDS)     # as the host parser declares it
    flags = torch.from_numpy(flags_np.reshape(-1).view(np.int32)).to(device)
    return coef, flags


# --------------------------------------------------------------- Vorbis ------

def vorbis_floor_curve(n2):
    k = np.arange(n2, dtype=np.float64)
    curve = VORBIS_LEVEL * 10.0 ** (-(k / n2) * 2.5) # ~50 dB down at Nyquist
    curve[int(n2 * 16000 / 22050):] = 0.0            # zeros above 16 kHz
    return curve.astype(np.float32)


def vorbis_packet_flags(rng, npkt, p_short_run=0.05):
    """Legal blockflag sequence with prev/next window flags (stb_vorbis2.d:2324-2331)."""
    long_ = np.ones(npkt, bool)
    i = 0
    while i < npkt:
        if rng.random() < p_short_run:
            run = int(rng.integers(1, 9))
            long_[i:i + run] = False
            i += run
        i += 1
    pf = np.zeros(npkt, np.uint8)
    for p in range(npkt):
        if long_[p]:
            prev_long = long_[p - 1] if p > 0 else True
            next_long = long_[p + 1] if p + 1 < npkt else True
            pf[p] = VORBIS_LONG | (VORBIS_PREV if prev_long else 0) | (VORBIS_NEXT if next_long else 0)
    return pf


def vorbis_batch(seed, packets, channels, bs0, bs1, p_short_run=0.05, amplitude=1.0):
    """numpy batch: returns (pflags[total], spec[spec_floats] f32).  Spectra are laid out
    packet after packet as [ch][n/2]."""
    pflags, specs = [], []
    for s, (npk, nc) in enumerate(zip(packets, channels)):
        rng = np.random.default_rng([seed, s])
        pf = vorbis_packet_flags(rng, int(npk), p_short_run)
        pflags.append(pf)
        for p in range(int(npk)):
            n2 = (int(bs1[s]) if (pf[p] & VORBIS_LONG) else int(bs0[s])) // 2
            x = rng.standard_normal((int(nc), n2)).astype(np.float32) * vorbis_floor_curve(n2)
            specs.append((x * np.float32(amplitude)).reshape(-1))
    if not specs:
        return np.zeros(0, np.uint8), np.zeros(0, np.float32)
    return np.concatenate(pflags), np.concatenate(specs)


# ----------------------------------------------------------------- FLAC ------

def _quantised_lpc(rng, order, precision=12):
    """A stable AR(order) predictor quantised like a FLAC encoder would (coef, shift)."""
    # poles inside the unit circle -> stable synthesis filter
    npairs = order // 2
    poles = []
    for _ in range(npairs):
        r = rng.uniform(0.5, 0.97)
        th = rng.uniform(0.02, np.pi * 0.9)
        poles += [r * np.exp(1j * th), r * np.exp(-1j * th)]
    if order % 2:
        poles.append(rng.uniform(-0.9, 0.9))
    a = np.real(np.poly(poles))                      # 1 + a1 z^-1 + ...
    lpc = -a[1:]                                      # prediction coefficients
    cmax = np.abs(lpc).max()
    shift = precision - 1 - int(np.floor(np.log2(cmax))) - 1
    shift = int(np.clip(shift, 0, 15))
    q = np.clip(np.round(lpc * (1 << shift)), -(1 << (precision - 1)), (1 << (precision - 1)) - 1)
    return q.astype(np.int16), shift


FLAC_C4_RESIDUAL_CLAMP = 1023      # C4 / C5 residuals: Laplacian of scale 2^5 (Rice k ~ 5), clamped here
FLAC_C4_L1_MAX = 24.0              # bound on sum |h| of a C4 predictor's synthesis filter 1 / A(z)


def stable_quantised_lpc(rng, order, precision=12, l1_max=FLAC_C4_L1_MAX):
    """A quantised AR(order) predictor as SURVEY 8d specifies it for C4: STABLE after quantisation, with a bounded gain --
    the impulse response h of the synthesis filter 1 / A_q(z) satisfies sum |h| <= l1_max, so residuals of magnitude
    <= R decode to samples of magnitude <= l1_max * R (+ the floor-shift rounding, < l1_max): with R = 1023 every sample
    stays inside 16 bits per subframe, 17 after mid / side decorrelation.  (Round 1-3's pool put pole pairs at up to
    0.97: a quarter of its filters were unstable once quantised and their samples wrapped around int32.)"""
    for _ in range(1000):
        npairs = order // 2
        poles = []
        for _ in range(npairs):
            r = rng.uniform(0.3, 0.9)
            th = rng.uniform(0.05, np.pi * 0.9)
            poles += [r * np.exp(1j * th), r * np.exp(-1j * th)]
        if order % 2:
            poles.append(rng.uniform(-0.8, 0.8))
        lpc = -np.real(np.poly(poles))[1:]
        cmax = np.abs(lpc).max()
        shift = int(np.clip(precision - 1 - int(np.floor(np.log2(cmax))) - 1, 0, 15))
        q = np.clip(np.round(lpc * (1 << shift)), -(1 << (precision - 1)), (1 << (precision - 1)) - 1).astype(np.int64)
        # impulse response of s[t] = d[t] + sum_k (q[k] / 2^shift) s[t-1-k]
        h = np.zeros(2048)
        h[0] = 1.0
        a = q / float(1 << shift)
        for t in range(1, len(h)):
            k = min(t, order)
            h[t] = np.dot(a[:k], h[t - 1::-1][:k])
        if np.abs(h).sum() <= l1_max and np.abs(h[-256:]).max() < 1e-6:
            return q.astype(np.int16), shift
    raise RuntimeError("no stable predictor found")


def flac_batch(seed, n_frames, block_size=4096, channels=2, bps=16, orders=(8, 12),
               assignments=(FLAC_MID_SIDE, FLAC_LEFT_SIDE, FLAC_RIGHT_SIDE, FLAC_INDEPENDENT),
               assignment_p=(0.55, 0.2, 0.1, 0.15), residual_scale=32.0, wasted_p=0.05,
               vary_block=False):
    """numpy batch of LPC subframes.  Returns (frames, subframes, res int32, out_total)."""
    rng = np.random.default_rng(seed)
    frames = np.zeros(n_frames, FLAC_FRAME_DTYPE)
    subframes = np.zeros(n_frames * channels, FLAC_SUBFRAME_DTYPE)
    res_parts = []
    in_off = out_off = 0
    for f in range(n_frames):
        bs = int(block_size if not vary_block else rng.choice([192, 576, 1152, 4096, 4608, 1000]))
        asg = int(rng.choice(assignments, p=assignment_p)) if channels == 2 else FLAC_INDEPENDENT
        frames[f] = (in_off, out_off, bs, f * channels, channels, asg, bps, 0, [0] * 4)
        for c in range(channels):
            side = (asg in (FLAC_LEFT_SIDE, FLAC_MID_SIDE) and c == 1) or (asg == FLAC_RIGHT_SIDE and c == 0)
            wasted = int(rng.integers(1, 3)) if rng.random() < wasted_p else 0
            sf_bps = bps + (1 if side else 0) - wasted
            order = int(rng.choice(orders))
            order = min(order, bs)
            if order > 0:
                coef, shift = _quantised_lpc(rng, order)
            else:
                coef, shift = np.zeros(0, np.int16), 0
            sf = subframes[f * channels + c]
            sf["coef"][:order] = coef
            sf["order"], sf["shift"], sf["wasted"], sf["use64"] = order, shift, wasted, int(sf_bps > 16)
            r = np.rint(rng.laplace(0.0, residual_scale, bs)).astype(np.int64)
            lim = (1 << (sf_bps - 1)) - 1
            r[:order] = rng.integers(-lim // 4, lim // 4 + 1, order)       # warm-up samples
            res_parts.append(r.astype(np.int32))
        in_off += bs * channels
        out_off += bs * channels
    return frames, subframes, np.concatenate(res_parts), out_off


def flac_pack16(frames, res, every=1):
    """The int16 storage of residual rows (AFG_FLAC_ROW16, include/afg.h): every `every`-th frame whose values all fit
    16 bits is rewritten as int16 rows padded to 8, the others stay int32; returns (frames, res int32 array holding both)."""
    frames = frames.copy()
    out = []
    words = 0
    for f in range(len(frames)):
        fr = frames[f]
        bs, C, off = int(fr["block_size"]), int(fr["channels"]), int(fr["in_off"])
        plane = res[off:off + bs * C]
        if f % every == 0 and bs >= 8 and np.abs(plane.astype(np.int64)).max(initial=0) < 32768:
            row = (bs + 7) & ~7
            packed = np.zeros(C * row, np.int16)
            for c in range(C):
                packed[c * row:c * row + bs] = plane[c * bs:(c + 1) * bs]
            pad = (-words) % 4                                    # 16-byte aligned start: in_off a multiple of 8 int16
            out.append(np.zeros(pad, np.int32))
            words += pad
            frames["in_off"][f] = 2 * words
            frames["res16"][f] = 1
            w = np.zeros((C * row + 1) // 2, np.int32)
            w.view(np.int16)[:C * row] = packed
            out.append(w)
            words += len(w)
        else:
            frames["in_off"][f] = words
            out.append(plane.astype(np.int32))
            words += bs * C
    return frames, (np.concatenate(out) if out else np.zeros(0, np.int32))


# ------------------------------------------------- device-resident BASELINE workloads ------

def vorbis_batch_device(seed, n_files, packets_per_file, device, bs0=256, bs1=2048, channels=2,
                        p_short_run=0.02, files_per_chunk=32):
    """C3-shaped batch built in HBM.  All files share one packet-flag sequence per seed%16 family
    so that spectra can be generated as dense tensors; returns (pflags_all, spec CUDA tensor, plan args)."""
    import torch
    from . import VorbisPlan
    fams = []
    for f in range(16):
        rng = np.random.default_rng([seed, f])
        fams.append(vorbis_packet_flags(rng, packets_per_file, p_short_run))
    pflags = np.concatenate([fams[f % 16] for f in range(n_files)])
    packets = np.full(n_files, packets_per_file, np.uint32)
    plan = VorbisPlan(packets, np.full(n_files, channels, np.uint8), np.full(n_files, bs0, np.uint16),
                      np.full(n_files, bs1, np.uint16), pflags, 0)
    spec = torch.empty(plan.spec_floats, dtype=torch.float32, device=device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    chunk = 1 << 28
    for o in range(0, plan.spec_floats, chunk):
        spec[o:o + chunk].normal_(generator=gen)
    # spectral shape: apply the long-block floor curve to long packets (dense majority); short packets keep N(0,1)*0.1
    so, _ = plan.offsets()
    curve = torch.from_numpy(vorbis_floor_curve(bs1 // 2)).to(device)
    is_long = (pflags & VORBIS_LONG) != 0
    if is_long.all():
        spec.view(-1, bs1 // 2).mul_(curve)
    else:
        spec.mul_(0.25 * VORBIS_LEVEL)
    return plan, spec


def flac_batch_device(seed, n_files, frames_per_file, device, block_size=4096, bps=16):
    """C4-shaped batch built in HBM: stereo, LPC order 8 (even files) / 12 (odd files), Laplacian
    residuals (scale 2^5), 60 % MID_SIDE / 25 % LEFT_SIDE / 15 % independent.  Returns
    (frames u8 tensor, subframes u8 tensor, res int32 tensor, n_frames, out_total)."""
    import torch
    rng = np.random.default_rng(seed)
    n_frames = n_files * frames_per_file
    pool = {o: [_quantised_lpc(np.random.default_rng([seed, o, i]), o) for i in range(64)] for o in (8, 12)}
    frames = np.zeros(n_frames, FLAC_FRAME_DTYPE)
    idx = np.arange(n_frames, dtype=np.uint64)
    frames["in_off"] = idx * np.uint64(block_size * 2)
    frames["out_off"] = idx * np.uint64(block_size * 2)
    frames["block_size"] = block_size
    frames["sf_index"] = (idx * 2).astype(np.uint32)
    frames["channels"] = 2
    frames["bps"] = bps
    asg = rng.choice([FLAC_MID_SIDE, FLAC_LEFT_SIDE, FLAC_INDEPENDENT], size=n_frames, p=[0.6, 0.25, 0.15])
    frames["assignment"] = asg.astype(np.uint8)
    subframes = np.zeros(n_frames * 2, FLAC_SUBFRAME_DTYPE)
    file_of = (idx // np.uint64(frames_per_file)).astype(np.int64)
    for order in (8, 12):
        sel_frames = np.flatnonzero((file_of % 2) == (0 if order == 8 else 1))
        for c in range(2):
            sidx = sel_frames * 2 + c
            pick = rng.integers(0, 64, sidx.size)
            coefs = np.stack([pool[order][i][0] for i in range(64)])        # [64, order]
            shifts = np.array([pool[order][i][1] for i in range(64)], np.uint8)
            subframes["coef"][sidx, :order] = coefs[pick]
            subframes["order"][sidx] = order
            subframes["shift"][sidx] = shifts[pick]
    side = np.zeros(n_frames * 2, bool)
    side[1::2] = np.isin(asg, [FLAC_MID_SIDE, FLAC_LEFT_SIDE])
    subframes["use64"] = side.astype(np.uint8)                                # subframe bps 17 > 16
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    total = n_frames * 2 * block_size
    res = torch.empty(total, dtype=torch.int32, device=device)
    chunk = 1 << 28
    for o in range(0, total, chunk):
        n = min(chunk, total - o)
        e = torch.empty(n, dtype=torch.float32, device=device).exponential_(1.0 / 32.0, generator=gen)
        sgn = torch.empty(n, dtype=torch.float32, device=device).uniform_(-1.0, 1.0, generator=gen).sign_()
        res[o:o + n] = (e * sgn).round_().to(torch.int32)
        del e, sgn
    d_frames = torch.from_numpy(frames.view(np.uint8).copy()).to(device)
    d_sub = torch.from_numpy(subframes.view(np.uint8).copy()).to(device)
    return d_frames, d_sub, res, n_frames, total, frames, subframes


# ----------------------------------------------------------------- CELT ------

def celt_batch(seed, frames_per_stream, channels, p_transient=0.15, p_postfilter=0.3, frame_sizes=(960,)):
    """numpy batch of CELT transform-stage records.  One channel sequence per (stream, channel);
    output interleaved per stream ([frame][sample][channel]).  Returns (rec_base, recs, coeffs, out_total)."""
    taps = np.array([[0.3066406250, 0.2170410156, 0.1296386719], [0.4638671875, 0.2680664062, 0.0],
                     [0.7998046875, 0.1000976562, 0.0]], np.float32)            # dopus.d:3382-3386
    recs, rec_base, coefs = [], [0], []
    coef_off = out_base = 0
    for s, (nf, C) in enumerate(zip(frames_per_stream, channels)):
        rng = np.random.default_rng([seed, s])
        sizes = rng.choice(frame_sizes, nf)
        trans = rng.random(nf) < p_transient
        haspf = rng.random(nf) < p_postfilter
        period = rng.integers(15, 1023, nf)
        gain = (0.09375 * (rng.integers(0, 8, nf) + 1)).astype(np.float32)      # dopus.d:3400
        tapset = rng.integers(0, 3, nf)
        starts = np.concatenate([[0], np.cumsum(sizes)])
        last_period = 0
        per_frame = []
        for f in range(nf):
            fs = int(sizes[f])
            blocks = (fs // 120) if trans[f] and fs > 120 else 1
            if haspf[f]:
                last_period = int(period[f])
                g = (gain[f] * taps[tapset[f]]).astype(np.float32)
            else:
                g = np.zeros(3, np.float32)                                     # gains reset, period kept (:3391)
            per_frame.append((fs, blocks, last_period, g))
        for c in range(C):
            for f in range(nf):
                fs, blocks, per, g = per_frame[f]
                k = np.arange(fs, dtype=np.float64)
                x = rng.standard_normal(fs) * 2000.0 * 10.0 ** (-(k / fs) * 2.0)
                coefs.append(x.astype(np.float32))
                recs.append((coef_off, out_base + int(starts[f]) * C + c, C, fs, blocks, 0, per, tuple(g), 1.0, 0))
                coef_off += fs
            rec_base.append(len(recs))
        out_base += int(starts[-1]) * C
    return (np.array(rec_base, np.uint64), np.array(recs, CELT_FRAME_DTYPE),
            np.concatenate(coefs) if coefs else np.zeros(0, np.float32), out_base)
