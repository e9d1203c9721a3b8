"""Vorbis transform stage: HIP path vs the CPU oracle (bit-exact float32), through the C ABI."""
import numpy as np
import pytest

import oraclelib
from afgpu import VorbisPlan, synthetic, VORBIS_LONG, VORBIS_PREV, VORBIS_NEXT

pytestmark = pytest.mark.gpu


def run_both(gpu, packets, channels, bs0, bs1, pflags, spec, seg=0):
    import torch
    plan = VorbisPlan(packets, channels, bs0, bs1, pflags, seg)
    spec_off, out_off = plan.offsets()
    o_spec_off, o_out_off, o_spec_total, o_out_total = oraclelib.vorbis_layout(packets, channels, bs0, bs1, pflags)
    # host layout logic of the product == oracle's restatement of stb_vorbis2.d:2333-2349 / :2645-2656
    assert (spec_off == o_spec_off).all() and (out_off == o_out_off).all()
    assert plan.spec_floats == o_spec_total == spec.size and plan.out_floats == o_out_total
    want = oraclelib.vorbis_transform(packets, channels, bs0, bs1, pflags, spec_off, out_off, spec, plan.out_floats)
    d_spec = torch.from_numpy(spec).to(gpu)
    d_out = torch.full((max(plan.out_floats, 1),), float("nan"), dtype=torch.float32, device=gpu)
    plan.transform(d_spec, d_out)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()[:plan.out_floats]
    return got, want


def compare(got, want):
    assert not np.isnan(got).any(), "unwritten output"
    diff = np.abs(got.astype(np.float64) - want.astype(np.float64))
    rms = float(np.sqrt(np.mean(diff ** 2))) if diff.size else 0.0
    assert rms <= 1e-5
    return int((got.view(np.uint32) != want.view(np.uint32)).sum())


@pytest.mark.parametrize("seg", [1, 2, 5, 16, 1000])
def test_vorbis_long_short_mix(gpu, seg):
    packets = [40, 1, 2, 25]
    channels = [2, 2, 1, 2]
    bs0 = [256] * 4
    bs1 = [2048] * 4
    pflags, spec = synthetic.vorbis_batch(7, packets, channels, bs0, bs1, p_short_run=0.2)
    got, want = run_both(gpu, packets, channels, bs0, bs1, pflags, spec, seg)
    assert compare(got, want) == 0


@pytest.mark.parametrize("bs", [(256, 256), (256, 512), (512, 1024), (1024, 4096), (2048, 8192), (512, 2048), (1024, 2048), (2048, 2048)])
def test_vorbis_blocksizes(gpu, bs):
    packets = [9, 6]
    channels = [2, 1]
    bs0 = [bs[0]] * 2
    bs1 = [bs[1]] * 2
    pflags, spec = synthetic.vorbis_batch(13, packets, channels, bs0, bs1, p_short_run=0.3)
    got, want = run_both(gpu, packets, channels, bs0, bs1, pflags, spec, 4)
    assert compare(got, want) == 0


def test_vorbis_multichannel_and_empty(gpu):
    packets = [12, 0, 7]
    channels = [6, 2, 3]
    bs0 = [256, 256, 512]
    bs1 = [1024, 2048, 2048]
    pflags, spec = synthetic.vorbis_batch(3, packets, channels, bs0, bs1, p_short_run=0.25)
    got, want = run_both(gpu, packets, channels, bs0, bs1, pflags, spec, 3)
    assert compare(got, want) == 0


def test_vorbis_many_channels_long_blocks(gpu):
    """16 channels of 2048-sample blocks, 8 of 8192: round 1's general kernel kept all channels of a segment in one
    workgroup's LDS and refused these; one wavefront per channel takes them."""
    packets = [7, 5, 6]
    channels = [16, 8, 3]
    bs0 = [256, 256, 2048]
    bs1 = [2048, 8192, 8192]
    pflags, spec = synthetic.vorbis_batch(31, packets, channels, bs0, bs1, p_short_run=0.25)
    got, want = run_both(gpu, packets, channels, bs0, bs1, pflags, spec, 3)
    assert compare(got, want) == 0


def test_vorbis_rejects_broken_reference_sizes(gpu):
    import afgpu
    with pytest.raises(afgpu.AfgError):
        VorbisPlan([4], [2], [64], [2048], np.full(4, VORBIS_LONG | VORBIS_PREV | VORBIS_NEXT, np.uint8))
    with pytest.raises(afgpu.AfgError):          # inconsistent window flags
        VorbisPlan([2], [2], [256], [2048], np.array([VORBIS_LONG | VORBIS_PREV | VORBIS_NEXT, 0], np.uint8))


def test_vorbis_tdac_reconstruction(gpu):
    """Size-independent property: forward MDCT with the Vorbis window then this path gives the
    signal back (time-domain alias cancellation), long blocks only."""
    n, npk = 2048, 8
    rng = np.random.default_rng(1)
    sig = rng.standard_normal((npk + 1) * (n // 2)).astype(np.float64)
    w = oraclelib.vorbis_tables(n)["window"].astype(np.float64)
    win = np.concatenate([w, w[::-1]])
    m = np.arange(n)[:, None]
    k = np.arange(n // 2)[None, :]
    basis = np.cos(np.pi / (2 * n) * (2 * m + 1 + n / 2) * (2 * k + 1))
    spec = np.zeros((npk, n // 2), np.float32)
    for p in range(npk):
        blk = sig[p * n // 2:p * n // 2 + n] * win
        spec[p] = (blk @ basis * (2.0 / (n // 2))).astype(np.float32)   # inverse is y = sum X cos(...), scale 1
    pflags = np.full(npk, VORBIS_LONG | VORBIS_PREV | VORBIS_NEXT, np.uint8)
    got, want = run_both(gpu, [npk], [1], [256], [n], pflags, spec.reshape(-1), 3)
    assert compare(got, want) == 0
    ref = sig[n // 2:n // 2 + got.size]
    assert np.abs(got - ref).max() < 2e-4
