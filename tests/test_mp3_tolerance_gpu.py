"""MP3 transform stage in the default numeric mode (AFG_NUMERIC_TOLERANCE): csrc/mp3_kernel.h compiled with fused
multiply-adds and the polyphase window (minimp3.d:1371-1405) accumulating in fma chains (csrc/mp3_tolerance.hip), against
the CPU oracle within north_star's 1e-5 RMS.  The twin of test_mp3_gpu.py; what must not depend on the arithmetic -- the
segmentation of a stream, chunked decoding through the state blob, declared-empty subbands -- is still compared bit for bit
against the kernel's own unsegmented result."""
import numpy as np
import pytest

import oraclelib
from afgpu import MP3_STATE_FLOATS, synthetic
from test_mp3_gpu import run_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.numeric_tolerance]


def check(got, want):
    assert got.shape == want.shape and not np.isnan(got).any(), "unwritten PCM"
    diff = got.astype(np.float64) - want.astype(np.float64)
    rms = float(np.sqrt(np.mean(diff ** 2))) if diff.size else 0.0
    sig = float(np.sqrt(np.mean(want.astype(np.float64) ** 2))) if diff.size else 0.0
    assert rms <= 1e-5 and rms <= 1e-5 * max(sig, 1e-3) * 100, (rms, sig)          # 1e-5 absolute; ~1e-6 of the signal measured
    return rms, int((got.view(np.uint32) != want.view(np.uint32)).sum())


def test_tolerance_kernel_is_the_path_under_test(gpu):
    granules, channels = [40], [2]
    coef, flags = synthetic.mp3_batch(11, granules, channels)
    rms, differ = check(run_gpu(gpu, granules, channels, coef, flags), oraclelib.mp3_transform(granules, channels, coef, flags))
    assert differ > 1000, "the exact kernel ran in the default mode"


@pytest.mark.parametrize("seg", [1, 2, 3, 7, 48, 1000])
def test_mp3_stereo_segmentation(gpu, seg):
    granules = [37, 1, 2, 3, 64]
    channels = [2, 2, 2, 2, 2]
    coef, flags = synthetic.mp3_batch(11, granules, channels)
    want = oraclelib.mp3_transform(granules, channels, coef, flags)
    got = run_gpu(gpu, granules, channels, coef, flags, seg)
    check(got, want)
    whole = run_gpu(gpu, granules, channels, coef, flags, 100000)
    assert (got.view(np.uint32) == whole.view(np.uint32)).all(), "the segment length changed the samples"


def test_mp3_mono_mixed_and_every_block_type(gpu):
    granules = [20, 33, 0, 5, 17]
    channels = [1, 2, 2, 1, 1]
    coef, flags = synthetic.mp3_batch(5, granules, channels, p_event=0.15, p_mixed=0.5)
    check(run_gpu(gpu, granules, channels, coef, flags, 6), oraclelib.mp3_transform(granules, channels, coef, flags))
    granules, channels = [40], [2]
    coef, _ = synthetic.mp3_batch(3, granules, channels)
    bt = np.array([1, 2, 2, 3, 0, 1, 2, 3] * 10, np.uint8)[:40]
    mixed = np.array([False, True, False, False, False, False, True, False] * 10)[:40]
    fl = synthetic.mp3_flag_words(bt, mixed)
    flags = np.stack([fl, np.roll(fl, 3)], 1).reshape(-1).astype(np.uint32)
    check(run_gpu(gpu, granules, channels, coef, flags, 5), oraclelib.mp3_transform(granules, channels, coef, flags))


def test_mp3_chunked_decode_with_state_equals_whole(gpu):
    granules, channels = [50, 31], [2, 1]
    coef, flags = synthetic.mp3_batch(9, granules, channels, p_event=0.1)
    want = oraclelib.mp3_transform(granules, channels, coef, flags)
    whole = run_gpu(gpu, granules, channels, coef, flags, 4)
    cuts = [23, 10]
    state = np.zeros((2, MP3_STATE_FLOATS), np.float32)
    out = np.zeros_like(want)
    blk_base = np.concatenate([[0], np.cumsum(np.array(granules) * np.array(channels))])
    for part in range(2):
        g_lo = [0, 0] if part == 0 else cuts
        g_hi = cuts if part == 0 else granules
        sel = np.concatenate([np.arange(blk_base[s] + g_lo[s] * channels[s], blk_base[s] + g_hi[s] * channels[s]) for s in range(2)])
        got, state = run_gpu(gpu, [g_hi[s] - g_lo[s] for s in range(2)], channels, coef.reshape(-1, 576)[sel].reshape(-1), flags[sel], 4,
                             state.reshape(-1))
        state = state.reshape(2, MP3_STATE_FLOATS)
        out.reshape(-1, 576)[sel] = got.reshape(-1, 576)
    check(out, want)
    assert (out.view(np.uint32) == whole.view(np.uint32)).all(), "chunked decoding through the state blob changed the samples"


def test_mp3_linearity_silence_and_declared_empty_subbands(gpu):
    granules, channels = [12], [2]
    coef, flags = synthetic.mp3_batch(21, granules, channels, p_event=0.0)
    assert (run_gpu(gpu, granules, channels, np.zeros_like(coef), flags) == 0).all()
    a = run_gpu(gpu, granules, channels, coef, flags)
    b = run_gpu(gpu, granules, channels, coef * np.float32(2.0), flags)
    assert (b == a * np.float32(2.0)).all()       # scaling by 2 is exact in float32, fused or not
    granules, channels = [70, 9, 131], [1, 2, 1]
    coef, flags = synthetic.mp3_batch(23, granules, channels, p_event=0.1, p_mixed=0.4)
    rng = np.random.default_rng(1)
    blocks = coef.reshape(-1, 576)
    nz = rng.integers(0, 33, len(blocks))
    for blk, n in zip(blocks, nz):
        blk[18 * n:] = 0.0
    want = oraclelib.mp3_transform(granules, channels, coef, flags)
    declared = flags | ((nz.astype(np.uint32) + 1) << 24)
    plain = run_gpu(gpu, granules, channels, coef, flags, 48)
    got = run_gpu(gpu, granules, channels, coef, declared, 5)
    check(got, want)
    assert (got.view(np.uint32) == plain.view(np.uint32)).all(), "declaring the empty subbands changed the samples"
