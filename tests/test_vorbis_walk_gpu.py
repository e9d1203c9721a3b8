"""Vorbis transform stage in the default numeric mode (AFG_NUMERIC_TOLERANCE): the re-factored walk of
csrc/vorbis_walk.hip (one 512-point complex FFT per 2048-sample block, window / overlap written on the DCT-IV) against
the CPU oracle's restatement of stb_vorbis' inverse_mdct (stb_vorbis2.d:1941-2242) + vorbis_finish_frame (:2606-2657).
The twin of test_vorbis_gpu.py: same cases, north_star's tolerance (1e-5 RMS on the API scale) instead of bit-identity,
plus checks that the result does not depend on how a stream is cut into segments."""
import numpy as np
import pytest

import oraclelib
from afgpu import VorbisPlan, synthetic, VORBIS_LONG, VORBIS_PREV, VORBIS_NEXT

pytestmark = [pytest.mark.gpu, pytest.mark.numeric_tolerance]

FULL = VORBIS_LONG | VORBIS_PREV | VORBIS_NEXT


def run_both(gpu, packets, channels, bs0, bs1, pflags, spec, seg=0):
    import torch
    plan = VorbisPlan(packets, channels, bs0, bs1, pflags, seg)
    spec_off, out_off = plan.offsets()
    want = oraclelib.vorbis_transform(packets, channels, bs0, bs1, pflags, spec_off, out_off, spec, plan.out_floats)
    d_spec = torch.from_numpy(spec).to(gpu)
    d_out = torch.full((max(plan.out_floats, 1),), float("nan"), dtype=torch.float32, device=gpu)
    plan.transform(d_spec, d_out)
    torch.cuda.synchronize()
    return d_out.cpu().numpy()[:plan.out_floats], want


def check(got, want, limit=1e-5):
    assert not np.isnan(got).any(), "unwritten output"
    diff = got.astype(np.float64) - want.astype(np.float64)
    rms = float(np.sqrt(np.mean(diff ** 2))) if diff.size else 0.0
    assert rms <= limit, f"rms error {rms} over the tolerance {limit}"
    sig = float(np.sqrt(np.mean(want.astype(np.float64) ** 2))) if want.size else 0.0
    assert sig == 0 or rms <= 1e-6 * max(sig, 1.0), f"rms error {rms} for a signal of rms {sig}"
    return rms, int((got.view(np.uint32) != want.view(np.uint32)).sum())


def test_walk_is_the_path_under_test(gpu):
    """The default mode must not be bit-identical on long blocks (it is a different factorisation); if it were, the exact
    kernel ran and this file would test nothing."""
    packets, channels = [12], [2]
    pflags = np.full(12, FULL, np.uint8)
    _, spec = synthetic.vorbis_batch(5, packets, channels, [256], [2048], p_short_run=0.0)
    got, want = run_both(gpu, packets, channels, [256], [2048], pflags, spec, 4)
    rms, differ = check(got, want)
    assert differ > got.size // 10, "tolerance mode produced the exact kernel's bits"


@pytest.mark.parametrize("seg", [1, 2, 5, 16, 1000])
def test_walk_long_short_mix(gpu, seg):
    packets = [40, 1, 2, 25, 70]
    channels = [2, 2, 1, 2, 2]
    bs0 = [256, 256, 256, 512, 256]
    bs1 = [2048] * 5
    pflags, spec = synthetic.vorbis_batch(7, packets, channels, bs0, bs1, p_short_run=0.2)
    got, want = run_both(gpu, packets, channels, bs0, bs1, pflags, spec, seg)
    check(got, want)


def test_walk_every_window_shape(gpu):
    """long after short, long before short, long between two shorts, short runs of every length, at both ends of a stream"""
    L, S = True, False
    shapes = [[L, L, L, L], [S, L, L, S], [L, S, L, S, L], [S, S, L, S, S], [L, L, S, S, S, L, L], [S] * 6, [L], [S],
              [L, S], [S, L], [L, L, S], [S, L, L, L, S, L, S, S, L]]
    packets, chans, flags = [], [], []
    for sh in shapes:
        n = len(sh)
        pf = np.zeros(n, np.uint8)
        for p in range(n):
            if sh[p]:
                prev_long = sh[p - 1] if p > 0 else True
                next_long = sh[p + 1] if p + 1 < n else True
                pf[p] = VORBIS_LONG | (VORBIS_PREV if prev_long else 0) | (VORBIS_NEXT if next_long else 0)
        packets.append(n)
        chans.append(2)
        flags.append(pf)
    pflags = np.concatenate(flags)
    rng = np.random.default_rng(11)
    spec = np.concatenate([rng.standard_normal(2 * (1024 if (f & VORBIS_LONG) else 128)).astype(np.float32) for f in pflags])
    for seg in (1, 3, 64):
        got, want = run_both(gpu, packets, chans, [256] * len(packets), [2048] * len(packets), pflags, spec, seg)
        check(got, want)


@pytest.mark.parametrize("bs0", [256, 512])
def test_walk_segmentation_does_not_change_the_result(gpu, bs0):
    packets, channels = [97, 33], [2, 2]
    pflags, spec = synthetic.vorbis_batch(21, packets, channels, [bs0] * 2, [2048] * 2, p_short_run=0.1)
    ref = None
    for seg in (1, 7, 16, 200):
        got, want = run_both(gpu, packets, channels, [bs0] * 2, [2048] * 2, pflags, spec, seg)
        check(got, want)
        if ref is None:
            ref = got
        else:
            assert (got.view(np.uint32) == ref.view(np.uint32)).all(), "the segment length changed the samples"


def test_walk_unaligned_planes_fall_back(gpu):
    """16-byte PCM stores need an aligned plane: a caller's odd offset takes the bit-exact kernel, not a fault."""
    import torch
    packets, channels = [9], [2]
    pflags = np.full(9, FULL, np.uint8)
    _, spec = synthetic.vorbis_batch(2, packets, channels, [256], [2048], p_short_run=0.0)
    plan = VorbisPlan(packets, channels, [256], [2048], pflags, 4)
    so, oo = plan.offsets()
    want = oraclelib.vorbis_transform(packets, channels, [256], [2048], pflags, so, oo, spec, plan.out_floats)
    d_spec = torch.from_numpy(spec).to(gpu)
    d_big = torch.full((plan.out_floats + 2,), float("nan"), dtype=torch.float32, device=gpu)
    plan.transform(d_spec, d_big[2:])                     # 8 bytes off
    torch.cuda.synchronize()
    got = d_big.cpu().numpy()[2:]
    assert (got.view(np.uint32) == want.view(np.uint32)).all()


def test_walk_tdac_reconstruction(gpu):
    """forward MDCT with the Vorbis window, then this path: the signal comes back (independent of the oracle)"""
    n, npk = 2048, 8
    rng = np.random.default_rng(1)
    sig = rng.standard_normal((npk + 1) * (n // 2)).astype(np.float64)
    w = oraclelib.vorbis_tables(n)["window"].astype(np.float64)
    win = np.concatenate([w, w[::-1]])
    m = np.arange(n)[:, None]
    k = np.arange(n // 2)[None, :]
    basis = np.cos(np.pi / (2 * n) * (2 * m + 1 + n / 2) * (2 * k + 1))
    spec = np.zeros((npk, 2, n // 2), np.float32)
    for p in range(npk):
        blk = sig[p * n // 2:p * n // 2 + n] * win
        spec[p, 0] = (blk @ basis * (2.0 / (n // 2))).astype(np.float32)
        spec[p, 1] = -spec[p, 0]
    pflags = np.full(npk, FULL, np.uint8)
    got, want = run_both(gpu, [npk], [2], [256], [n], pflags, spec.reshape(-1), 3)
    check(got, want)
    ref = sig[n // 2:n // 2 + got.size // 2]
    assert np.abs(got[0::2] - ref).max() < 2e-4 and np.abs(got[1::2] + ref).max() < 2e-4


def test_walk_amplitude_range(gpu):
    """programme level (rms 0.1) and 30 dB over full scale: the error scales with the signal"""
    packets, channels = [20], [2]
    for amp in (0.02, 30.0):
        pflags, spec = synthetic.vorbis_batch(4, packets, channels, [256], [2048], p_short_run=0.1, amplitude=amp)
        got, want = run_both(gpu, packets, channels, [256], [2048], pflags, spec, 6)
        diff = got.astype(np.float64) - want.astype(np.float64)
        rms, sig = np.sqrt(np.mean(diff ** 2)), np.sqrt(np.mean(want.astype(np.float64) ** 2))
        assert rms <= 1e-6 * sig, (amp, rms, sig)


@pytest.mark.parametrize("eighths", [0, 1, 5, 6, 8])
def test_walk_declared_zero_tail(gpu, eighths):
    """AFG_VORBIS_NZ_EIGHTHS (afg.h): a long packet whose flags say "only the first e eighths of the spectrum can be nonzero"
    decodes to the same bits as the same packet with nothing declared -- and what lies in the declared-empty part of the
    plane is never looked at (NaN there leaves no trace)."""
    from afgpu import VORBIS_NZ_EIGHTHS
    packets, channels = [31, 14], [2, 2]
    pflags, spec = synthetic.vorbis_batch(17 + eighths, packets, channels, [256] * 2, [2048] * 2, p_short_run=0.15)
    plan = VorbisPlan(packets, channels, [256] * 2, [2048] * 2, pflags, 5)
    so, _ = plan.offsets()
    spec = spec.copy()
    poisoned = spec.copy()
    for p, f in enumerate(pflags):
        if f & VORBIS_LONG:
            for c in range(2):
                o = int(so[p]) + 1024 * c
                spec[o + 128 * eighths:o + 1024] = 0.0
                poisoned[o + 128 * eighths:o + 1024] = np.nan
    plain, want = run_both(gpu, packets, channels, [256] * 2, [2048] * 2, pflags, spec, 5)
    check(plain, want)
    declared = np.where(pflags & VORBIS_LONG, pflags | VORBIS_NZ_EIGHTHS(eighths), pflags).astype(np.uint8)
    got, want2 = run_both(gpu, packets, channels, [256] * 2, [2048] * 2, declared, spec, 5)
    assert (want2.view(np.uint32) == want.view(np.uint32)).all(), "the oracle must ignore the declaration"
    assert (got.view(np.uint32) == plain.view(np.uint32)).all()
    import torch
    d_out = torch.full((plan.out_floats,), float("nan"), dtype=torch.float32, device=gpu)
    VorbisPlan(packets, channels, [256] * 2, [2048] * 2, declared, 5).transform(torch.from_numpy(poisoned).to(gpu), d_out)
    torch.cuda.synchronize()
    assert (d_out.cpu().numpy().view(np.uint32) == plain.view(np.uint32)).all()


# ---------------------------------------------------------------- round 5: mono streams, blocksize_1 = 1024 and 4096
SHAPES = [(1, 256, 2048), (1, 512, 2048), (2, 256, 1024), (1, 256, 1024), (2, 512, 1024), (2, 512, 4096), (1, 512, 4096),
          (2, 256, 4096), (1, 256, 4096),
          # more than two channels: a workgroup per segment, one wavefront per channel (3, 5, 7) or pair of channels (even counts),
          # whole interleaved frames put together in LDS
          (3, 256, 2048), (6, 256, 2048), (6, 512, 1024), (4, 512, 4096), (8, 256, 2048), (16, 256, 2048),
          (5, 256, 1024), (7, 256, 2048), (7, 512, 4096), (8, 512, 4096), (12, 256, 2048), (16, 256, 1024),
          # one block size (libvorbis' lowest 16 / 22 kHz modes): every packet a long block between long blocks
          (2, 1024, 1024), (1, 1024, 1024), (2, 2048, 2048), (3, 1024, 1024)]


def legal_flags(longs):
    n = len(longs)
    pf = np.zeros(n, np.uint8)
    for p in range(n):
        if longs[p]:
            prev_long = longs[p - 1] if p > 0 else True
            next_long = longs[p + 1] if p + 1 < n else True
            pf[p] = VORBIS_LONG | (VORBIS_PREV if prev_long else 0) | (VORBIS_NEXT if next_long else 0)
    return pf


@pytest.mark.parametrize("ch,bs0,bs1", SHAPES)
def test_walk_shape_is_the_path_under_test(gpu, ch, bs0, bs1):
    packets, channels = [12], [ch]
    pflags = np.full(12, FULL, np.uint8)
    _, spec = synthetic.vorbis_batch(5, packets, channels, [bs0], [bs1], p_short_run=0.0)
    got, want = run_both(gpu, packets, channels, [bs0], [bs1], pflags, spec, 4)
    rms, differ = check(got, want)
    assert differ > got.size // 10, "tolerance mode produced the exact kernel's bits"


@pytest.mark.parametrize("ch,bs0,bs1", [(9, 256, 2048), (10, 512, 4096), (15, 256, 1024)])
def test_channel_counts_the_walk_leaves_to_the_exact_kernels(gpu, ch, bs0, bs1):
    """an odd channel count above 7 and 4096-sample blocks of more than 8 channels (more wavefronts than a 512-thread
    workgroup has) stay on the bit-exact kernels in the default mode too"""
    packets = [14, 3]
    pflags, spec = synthetic.vorbis_batch(9, packets, [ch] * 2, [bs0] * 2, [bs1] * 2, p_short_run=0.2)
    got, want = run_both(gpu, packets, [ch] * 2, [bs0] * 2, [bs1] * 2, pflags, spec, 4)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()


@pytest.mark.parametrize("seg", [1, 5, 16, 1000])
@pytest.mark.parametrize("ch,bs0,bs1", SHAPES)
def test_walk_shape_long_short_mix(gpu, ch, bs0, bs1, seg):
    packets = [40, 1, 2, 25, 70]
    pflags, spec = synthetic.vorbis_batch(7, packets, [ch] * 5, [bs0] * 5, [bs1] * 5, p_short_run=0.2)
    got, want = run_both(gpu, packets, [ch] * 5, [bs0] * 5, [bs1] * 5, pflags, spec, seg)
    check(got, want)


@pytest.mark.parametrize("ch,bs0,bs1", SHAPES)
def test_walk_shape_every_window_shape(gpu, ch, bs0, bs1):
    L, S = True, False
    shapes = [[L, L, L, L], [S, L, L, S], [L, S, L, S, L], [S, S, L, S, S], [L, L, S, S, S, L, L], [S] * 6, [L], [S],
              [L, S], [S, L], [L, L, S], [S, L, L, L, S, L, S, S, L]]
    flags = [legal_flags(sh) for sh in shapes]
    packets = [len(sh) for sh in shapes]
    pflags = np.concatenate(flags)
    rng = np.random.default_rng(11)
    spec = np.concatenate([rng.standard_normal(ch * ((bs1 if (f & VORBIS_LONG) else bs0) // 2)).astype(np.float32) for f in pflags])
    k = len(packets)
    for seg in (1, 3, 64):
        got, want = run_both(gpu, packets, [ch] * k, [bs0] * k, [bs1] * k, pflags, spec, seg)
        check(got, want)


@pytest.mark.parametrize("ch,bs0,bs1", SHAPES)
def test_walk_shape_segmentation_does_not_change_the_result(gpu, ch, bs0, bs1):
    packets, channels = [97, 33], [ch, ch]
    pflags, spec = synthetic.vorbis_batch(21, packets, channels, [bs0] * 2, [bs1] * 2, p_short_run=0.1)
    ref = None
    for seg in (1, 7, 16, 200):
        got, want = run_both(gpu, packets, channels, [bs0] * 2, [bs1] * 2, pflags, spec, seg)
        check(got, want)
        if ref is None:
            ref = got
        else:
            assert (got.view(np.uint32) == ref.view(np.uint32)).all(), "the segment length changed the samples"


@pytest.mark.parametrize("ch,n", [(1, 2048), (1, 1024), (2, 1024), (1, 4096), (2, 4096)])
def test_walk_shape_tdac_reconstruction(gpu, ch, n):
    """forward MDCT with the Vorbis window, then this path: the signal comes back (independent of the oracle)"""
    npk = 8
    rng = np.random.default_rng(1)
    sig = rng.standard_normal((npk + 1) * (n // 2)).astype(np.float64)
    w = oraclelib.vorbis_tables(n)["window"].astype(np.float64)
    win = np.concatenate([w, w[::-1]])
    m = np.arange(n)[:, None]
    k = np.arange(n // 2)[None, :]
    basis = np.cos(np.pi / (2 * n) * (2 * m + 1 + n / 2) * (2 * k + 1))
    spec = np.zeros((npk, ch, n // 2), np.float32)
    for p in range(npk):
        blk = sig[p * n // 2:p * n // 2 + n] * win
        spec[p, 0] = (blk @ basis * (2.0 / (n // 2))).astype(np.float32)
        if ch == 2:
            spec[p, 1] = -spec[p, 0]
    pflags = np.full(npk, FULL, np.uint8)
    got, want = run_both(gpu, [npk], [ch], [256], [n], pflags, spec.reshape(-1), 3)
    check(got, want)
    ref = sig[n // 2:n // 2 + got.size // ch]
    assert np.abs(got[0::ch] - ref).max() < 4e-4
    if ch == 2:
        assert np.abs(got[1::2] + ref).max() < 4e-4


@pytest.mark.parametrize("eighths", [0, 1, 3, 6, 8])
@pytest.mark.parametrize("ch,bs0,bs1", [(1, 256, 2048), (2, 256, 1024), (1, 256, 1024), (2, 512, 4096), (1, 512, 4096),
                                        (6, 256, 2048), (3, 256, 1024)])
def test_walk_shape_declared_zero_tail(gpu, ch, bs0, bs1, eighths):
    from afgpu import VORBIS_NZ_EIGHTHS
    import torch
    packets, channels = [31, 14], [ch, ch]
    pflags, spec = synthetic.vorbis_batch(17 + eighths, packets, channels, [bs0] * 2, [bs1] * 2, p_short_run=0.15)
    plan = VorbisPlan(packets, channels, [bs0] * 2, [bs1] * 2, pflags, 5)
    so, _ = plan.offsets()
    spec = spec.copy()
    poisoned = spec.copy()
    n2 = bs1 // 2
    for p, f in enumerate(pflags):
        if f & VORBIS_LONG:
            for c in range(ch):
                o = int(so[p]) + n2 * c
                spec[o + n2 // 8 * eighths:o + n2] = 0.0
                poisoned[o + n2 // 8 * eighths:o + n2] = np.nan
    plain, want = run_both(gpu, packets, channels, [bs0] * 2, [bs1] * 2, pflags, spec, 5)
    check(plain, want)
    declared = np.where(pflags & VORBIS_LONG, pflags | VORBIS_NZ_EIGHTHS(eighths), pflags).astype(np.uint8)
    got, want2 = run_both(gpu, packets, channels, [bs0] * 2, [bs1] * 2, declared, spec, 5)
    assert (want2.view(np.uint32) == want.view(np.uint32)).all(), "the oracle must ignore the declaration"
    assert (got.view(np.uint32) == plain.view(np.uint32)).all()
    d_out = torch.full((plan.out_floats,), float("nan"), dtype=torch.float32, device=gpu)
    VorbisPlan(packets, channels, [bs0] * 2, [bs1] * 2, declared, 5).transform(torch.from_numpy(poisoned).to(gpu), d_out)
    torch.cuda.synchronize()
    assert (d_out.cpu().numpy().view(np.uint32) == plain.view(np.uint32)).all()


def test_walk_every_shape_in_one_plan(gpu):
    """streams of every shape the walk takes and of shapes it does not (8192-sample blocks, equal block sizes,
    blocksize_0 = 1024) in one plan: each goes to its own kernel and the planes are written once"""
    ch = [2, 1, 2, 1, 2, 1, 3, 2, 1, 2, 2, 1, 5, 6, 4]
    bs0 = [256, 256, 256, 512, 512, 256, 256, 256, 512, 1024, 512, 1024, 1024, 256, 256]
    bs1 = [2048, 2048, 1024, 1024, 4096, 4096, 2048, 8192, 512, 2048, 2048, 2048, 2048, 1024, 4096]
    packets = [23, 31, 40, 17, 19, 26, 9, 5, 30, 12, 44, 21, 8, 15, 11]
    pflags, spec = synthetic.vorbis_batch(77, packets, ch, bs0, bs1, p_short_run=0.15)
    for seg in (4, 16):
        got, want = run_both(gpu, packets, ch, bs0, bs1, pflags, spec, seg)
        check(got, want)
    # the exact kernels on the same plan (AFG_NUMERIC_EXACT): bit-identical
    import afgpu
    afgpu.set_numeric_mode(afgpu.NUMERIC_EXACT)
    try:
        got, want = run_both(gpu, packets, ch, bs0, bs1, pflags, spec, 16)
        assert (got.view(np.uint32) == want.view(np.uint32)).all()
    finally:
        afgpu.set_numeric_mode(afgpu.NUMERIC_FROM_ENV)
