"""MP3 transform stage: HIP path vs the CPU oracle (bit-exact float32), through the C ABI."""
import numpy as np
import pytest

import oraclelib
from afgpu import Mp3Plan, MP3_STATE_FLOATS, synthetic

pytestmark = pytest.mark.gpu


def run_gpu(gpu, granules, channels, coef, flags, seg=0, state=None):
    import torch
    plan = Mp3Plan(granules, channels, seg)
    assert plan.blocks * 576 == coef.size
    d_coef = torch.from_numpy(coef).to(gpu)
    d_flags = torch.from_numpy(flags.view(np.int32)).to(gpu)
    d_pcm = torch.full((coef.size,), float("nan"), dtype=torch.float32, device=gpu)
    d_state = None if state is None else torch.from_numpy(state).to(gpu)
    plan.transform(d_coef, d_flags, d_pcm, d_state)
    torch.cuda.synchronize()
    out = d_pcm.cpu().numpy()
    if state is not None:
        return out, d_state.cpu().numpy()
    return out


def compare(got, want):
    assert got.shape == want.shape
    assert not np.isnan(got).any(), "unwritten PCM"
    diff = np.abs(got.astype(np.float64) - want.astype(np.float64))
    rms = float(np.sqrt(np.mean(diff ** 2))) if diff.size else 0.0
    nbad = int((got.view(np.uint32) != want.view(np.uint32)).sum())
    assert rms <= 1e-5, f"rms {rms}"          # north_star tolerance
    return rms, nbad


@pytest.mark.parametrize("seg", [1, 2, 3, 7, 48, 1000])
def test_mp3_stereo_segmentation(gpu, seg):
    granules = [37, 1, 2, 3, 64]
    channels = [2, 2, 2, 2, 2]
    coef, flags = synthetic.mp3_batch(11, granules, channels)
    want = oraclelib.mp3_transform(granules, channels, coef, flags)
    got = run_gpu(gpu, granules, channels, coef, flags, seg)
    rms, nbad = compare(got, want)
    assert nbad == 0, f"{nbad} samples differ bitwise (rms {rms})"


def test_mp3_mono_and_mixed_streams(gpu):
    granules = [20, 33, 0, 5, 17]
    channels = [1, 2, 2, 1, 1]
    coef, flags = synthetic.mp3_batch(5, granules, channels, p_event=0.15, p_mixed=0.5)
    want = oraclelib.mp3_transform(granules, channels, coef, flags)
    got = run_gpu(gpu, granules, channels, coef, flags, 6)
    rms, nbad = compare(got, want)
    assert nbad == 0


def test_mp3_all_block_types_every_granule(gpu):
    # every channel cycles start/short/stop/mixed-short constantly
    granules = [40]
    channels = [2]
    coef, _ = synthetic.mp3_batch(3, granules, channels)
    bt = np.array([1, 2, 2, 3, 0, 1, 2, 3] * 10, np.uint8)[:40]
    mixed = np.array([False, True, False, False, False, False, True, False] * 10)[:40]
    fl = synthetic.mp3_flag_words(bt, mixed)
    flags = np.stack([fl, np.roll(fl, 3)], 1).reshape(-1).astype(np.uint32)
    want = oraclelib.mp3_transform(granules, channels, coef, flags)
    got = run_gpu(gpu, granules, channels, coef, flags, 5)
    rms, nbad = compare(got, want)
    assert nbad == 0


def test_mp3_chunked_decode_with_state_equals_whole(gpu):
    granules = [50, 31]
    channels = [2, 1]
    coef, flags = synthetic.mp3_batch(9, granules, channels, p_event=0.1)
    want = oraclelib.mp3_transform(granules, channels, coef, flags)
    # split each stream in two chunks, carrying the opaque state blob
    cuts = [23, 10]
    state = np.zeros((2, MP3_STATE_FLOATS), np.float32)
    out = np.zeros_like(want)
    blk_base = np.concatenate([[0], np.cumsum(np.array(granules) * np.array(channels))])
    for part in range(2):
        g_lo = [0, 0] if part == 0 else cuts
        g_hi = cuts if part == 0 else granules
        sel = []
        for s in range(2):
            b0 = blk_base[s] + g_lo[s] * channels[s]
            b1 = blk_base[s] + g_hi[s] * channels[s]
            sel.append(np.arange(b0, b1))
        sel = np.concatenate(sel)
        c = coef.reshape(-1, 576)[sel].reshape(-1)
        f = flags[sel]
        ng = [g_hi[s] - g_lo[s] for s in range(2)]
        got, state = run_gpu(gpu, ng, channels, c, f, 4, state.reshape(-1))
        state = state.reshape(2, MP3_STATE_FLOATS)
        out.reshape(-1, 576)[sel] = got.reshape(-1, 576)
    rms, nbad = compare(out, want)
    assert nbad == 0


def test_mp3_linearity_and_silence(gpu):
    granules = [12]
    channels = [2]
    coef, flags = synthetic.mp3_batch(21, granules, channels, p_event=0.0)
    z = run_gpu(gpu, granules, channels, np.zeros_like(coef), flags)
    assert (z == 0).all()
    a = run_gpu(gpu, granules, channels, coef, flags)
    b = run_gpu(gpu, granules, channels, coef * np.float32(2.0), flags)
    assert (b == a * np.float32(2.0)).all()       # scaling by 2 is exact in float32


@pytest.mark.parametrize("channels", [[2, 2, 2], [1, 2, 1]])
def test_mp3_declared_empty_subbands_are_not_needed(gpu, channels):
    """AFG_MP3_NZ_BANDS: with the tails declared the device may skip them -- same PCM as the oracle, which reads all
    576 lines -- including per-granule varying counts, 0 bands (silence) and all 32; and more than 64 granules, so the
    lane-register flag window is refilled."""
    granules = [70, 9, 131]
    coef, flags = synthetic.mp3_batch(23, granules, channels, p_event=0.1, p_mixed=0.4)
    rng = np.random.default_rng(1)
    blocks = coef.reshape(-1, 576)
    nz = rng.integers(0, 33, len(blocks))
    nz[::7] = 0
    nz[3::11] = 32
    for b, n in zip(blocks, nz):
        b[18 * n:] = 0.0
    want = oraclelib.mp3_transform(granules, channels, coef, flags)
    declared = flags | ((nz.astype(np.uint32) + 1) << 24)
    for seg in (5, 48):
        got = run_gpu(gpu, granules, channels, coef, declared, seg)
        rms, nbad = compare(got, want)
        assert nbad == 0, f"{nbad} samples differ bitwise (rms {rms})"
    # the contract: data in a subband declared empty is ignored (here garbage, even NaN)
    poisoned = blocks.copy()
    for b, n in zip(poisoned, nz):
        b[18 * n:] = np.nan
    got = run_gpu(gpu, granules, channels, poisoned.reshape(-1), declared, 48)
    assert compare(got, want)[1] == 0
