"""FLAC restore: HIP path vs the CPU oracle, bit-exact int32 and float, through the C ABI."""
import numpy as np
import pytest

import oraclelib
import afgpu
from afgpu import synthetic

pytestmark = pytest.mark.gpu


def run_gpu(gpu, frames, subframes, res, out_total, want_float=True):
    import torch
    d_frames = torch.from_numpy(frames.view(np.uint8).copy()).to(gpu)
    d_sub = torch.from_numpy(subframes.view(np.uint8).copy()).to(gpu)
    d_res = torch.from_numpy(res).to(gpu)
    d_i32 = torch.full((out_total,), -12345, dtype=torch.int32, device=gpu)
    d_f32 = torch.full((out_total,), float("nan"), dtype=torch.float32, device=gpu) if want_float else None
    afgpu.flac_transform(len(frames), d_frames, d_sub, d_res, d_i32, d_f32)
    torch.cuda.synchronize()
    return d_i32.cpu().numpy(), (d_f32.cpu().numpy() if want_float else None)


@pytest.mark.parametrize("kw", [
    dict(n_frames=130, block_size=4096, orders=(8, 12)),
    dict(n_frames=70, block_size=1152, orders=(0, 1, 2, 3, 4)),
    dict(n_frames=65, block_size=4096, orders=(13, 16, 24, 32)),
    dict(n_frames=200, vary_block=True, orders=(2, 8, 12, 31)),
    dict(n_frames=33, block_size=576, channels=1, orders=(8, 12)),
    dict(n_frames=20, block_size=500, channels=6, orders=(4, 8)),
    dict(n_frames=40, block_size=4096, bps=24, orders=(8, 12), residual_scale=3000.0),
    dict(n_frames=1, block_size=16, orders=(8,)),
])
def test_flac_bit_exact(gpu, kw):
    frames, subframes, res, total = synthetic.flac_batch(17, **kw)
    want_i, want_f = oraclelib.flac_transform(frames, subframes, res, total, want_float=True)
    got_i, got_f = run_gpu(gpu, frames, subframes, res, total)
    assert (got_i == want_i).all(), f"{int((got_i != want_i).sum())} int32 mismatches"
    assert (got_f.view(np.uint32) == want_f.view(np.uint32)).all()


def test_flac_wrapping_arithmetic(gpu):
    """Garbage-in must equal garbage-out: huge residuals/coefficients exercise the int32 wrap of
    drflac__calculate_prediction_32 and the int64 path of _64."""
    rng = np.random.default_rng(5)
    frames, subframes, res, total = synthetic.flac_batch(23, n_frames=96, block_size=1024, orders=(12, 32))
    subframes["coef"] = rng.integers(-32768, 32768, subframes["coef"].shape).astype(np.int16)
    for sf in subframes:
        sf["coef"][sf["order"]:] = 0
    subframes["shift"] = rng.integers(0, 32, len(subframes)).astype(np.uint8)
    subframes["use64"] = rng.integers(0, 2, len(subframes)).astype(np.uint8)
    res = rng.integers(-2**31, 2**31, res.shape, dtype=np.int64).astype(np.int32)
    want_i = oraclelib.flac_transform(frames, subframes, res, total)
    got_i, _ = run_gpu(gpu, frames, subframes, res, total, want_float=False)
    assert (got_i == want_i).all()


def test_flac_float_only_output(gpu):
    import torch
    frames, subframes, res, total = synthetic.flac_batch(29, n_frames=10, block_size=256)
    _, want_f = oraclelib.flac_transform(frames, subframes, res, total, want_float=True)
    d_frames = torch.from_numpy(frames.view(np.uint8).copy()).to(gpu)
    d_sub = torch.from_numpy(subframes.view(np.uint8).copy()).to(gpu)
    d_res = torch.from_numpy(res).to(gpu)
    d_f32 = torch.zeros(total, dtype=torch.float32, device=gpu)
    afgpu.flac_transform(len(frames), d_frames, d_sub, d_res, None, d_f32)
    torch.cuda.synchronize()
    assert (d_f32.cpu().numpy().view(np.uint32) == want_f.view(np.uint32)).all()
    with pytest.raises(afgpu.AfgError):
        afgpu.flac_transform(len(frames), d_frames, d_sub, d_res, None, None)


@pytest.mark.parametrize("kw,every", [
    (dict(n_frames=130, block_size=4096, orders=(8, 12)), 1),                 # every frame packed
    (dict(n_frames=150, block_size=4096, orders=(8, 12)), 3),                 # packed and unpacked frames inside one wavefront
    (dict(n_frames=200, vary_block=True, orders=(2, 8, 12, 31)), 2),           # odd block sizes: rows padded to 8
    (dict(n_frames=33, block_size=576, channels=1, orders=(8, 12)), 1),
    (dict(n_frames=20, block_size=500, channels=6, orders=(4, 8)), 1),
    (dict(n_frames=40, block_size=4096, bps=24, orders=(8, 12), residual_scale=3000.0), 1),   # hardly any frame fits
])
def test_flac_int16_residual_rows(gpu, kw, every):
    """afg_flac_frame.res16: the same samples from int16 rows as from the int32 planes"""
    frames, subframes, res, total = synthetic.flac_batch(23, **kw)
    want_i, want_f = oraclelib.flac_transform(frames, subframes, res, total, want_float=True)
    pf, pres = synthetic.flac_pack16(frames, res, every)
    assert pf["res16"].any() or kw.get("bps") == 24
    assert len(pres) < len(res) or not pf["res16"].any()
    oi, of = oraclelib.flac_transform(pf, subframes, pres, total, want_float=True)         # the oracle reads both forms alike
    assert (oi == want_i).all() and (of.view(np.uint32) == want_f.view(np.uint32)).all()
    got_i, got_f = run_gpu(gpu, pf, subframes, pres, total)
    assert (got_i == want_i).all(), f"{int((got_i != want_i).sum())} int32 mismatches"
    assert (got_f.view(np.uint32) == want_f.view(np.uint32)).all()


@pytest.mark.parametrize("kw", [
    dict(n_frames=130, block_size=4096, orders=(8, 12)),
    dict(n_frames=96, block_size=4096, channels=1, orders=(8, 12)),            # mono: slot B of every tile row stays empty
    dict(n_frames=70, block_size=1152, orders=(0, 1, 2, 3, 4)),
    dict(n_frames=65, block_size=4096, orders=(13, 16, 24, 32)),
    dict(n_frames=200, vary_block=True, orders=(2, 8, 12, 31)),                # ragged blocks: common and general steps alternate
    dict(n_frames=64, block_size=4096, bps=24, orders=(8, 12), residual_scale=3000.0),
    dict(n_frames=40, block_size=500, channels=6, orders=(4, 8)),              # (never the common step: stays on the general one)
])
@pytest.mark.parametrize("rows16", [False, True])
@pytest.mark.parametrize("out", ["int32", "float"])
def test_flac_common_step_one_output(gpu, kw, rows16, out):
    """The branch-free common step of the restore kernel (csrc/flac_restore.hip, round 5) runs when exactly ONE output is
    asked for -- what afg_batch_decode, the stream surface and bench.py do -- and every frame of a wavefront is stereo (or
    mono) with one row width; the tests above ask for both outputs and so take the general step.  Same samples either way."""
    import torch
    frames, subframes, res, total = synthetic.flac_batch(31, **kw)
    want_i, want_f = oraclelib.flac_transform(frames, subframes, res, total, want_float=True)
    if rows16:
        frames, res = synthetic.flac_pack16(frames, res, 1)
    d_frames = torch.from_numpy(frames.view(np.uint8).copy()).to(gpu)
    d_sub = torch.from_numpy(subframes.view(np.uint8).copy()).to(gpu)
    d_res = torch.from_numpy(res).to(gpu)
    if out == "int32":
        d_out = torch.full((total,), -12345, dtype=torch.int32, device=gpu)
        afgpu.flac_transform(len(frames), d_frames, d_sub, d_res, d_out, None)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        assert (got == want_i).all(), f"{int((got != want_i).sum())} int32 mismatches, first at {int(np.nonzero(got != want_i)[0][0])}"
    else:
        d_out = torch.full((total,), float("nan"), dtype=torch.float32, device=gpu)
        afgpu.flac_transform(len(frames), d_frames, d_sub, d_res, None, d_out)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        assert (got.view(np.uint32) == want_f.view(np.uint32)).all()


def test_flac_variant_mask_launches_the_same_samples(gpu):
    """afg_flac_variants on the host records + afg_flac_transform_variants_hip (only the populated instantiations, two
    streams) against the plain entry and the oracle: mixed orders and accumulator widths so that several instantiations
    are populated; a mask that leaves one out leaves exactly that instantiation's frames unwritten."""
    import torch
    groups = [((2, 3), 16), ((8, 6), 16), ((12, 9), 16), ((32, 20), 16), ((8, 8), 24), ((1, 12), 24)]
    frs, sfs, rss, in_off, sf_off = [], [], [], 0, 0
    for i, (orders, bps) in enumerate(groups):                 # 64 frames each: every group lands in one instantiation
        fr, sf, res, total = synthetic.flac_batch(770 + i, 64, block_size=192, orders=orders, bps=bps)
        fr["in_off"] += np.uint64(in_off)
        fr["out_off"] += np.uint64(in_off)
        fr["sf_index"] += np.uint32(sf_off)
        frs.append(fr); sfs.append(sf); rss.append(res)
        in_off += total
        sf_off += len(sf)
    frames, subs, res, total = np.concatenate(frs), np.concatenate(sfs), np.concatenate(rss), in_off
    mask = afgpu.flac_variants(frames, subs)
    assert bin(mask).count("1") >= 4
    want = oraclelib.flac_transform(frames, subs, res, total)
    d_fr = torch.from_numpy(frames.view(np.uint8).copy()).to(gpu)
    d_sf = torch.from_numpy(subs.view(np.uint8).copy()).to(gpu)
    d_res = torch.from_numpy(res).to(gpu)
    for variants in (None, mask, 0xffff):
        out = torch.full((total,), -7, dtype=torch.int32, device=gpu)
        afgpu.flac_transform(len(frames), d_fr, d_sf, d_res, out, None, None, variants=variants)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want)
    # drop the lowest populated bit: its frames keep the fill value, everything else is decoded
    low = mask & -mask
    out = torch.full((total,), -7, dtype=torch.int32, device=gpu)
    afgpu.flac_transform(len(frames), d_fr, d_sf, d_res, out, None, None, variants=mask & ~low)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    untouched = got == -7
    assert untouched.any() and np.array_equal(got[~untouched], want[~untouched])
    share = untouched.reshape(len(groups), -1).mean(axis=1)
    assert ((share < 0.01) | (share > 0.99)).all() and (share > 0.99).sum() >= 1           # whole groups or nothing


def test_flac_transform_is_stream_ordered(gpu):
    """afg_flac_transform_hip reads the records when the stream gets there, not at the call: records that earlier work on
    the stream is still producing (here: copied into page-locked memory behind a long kernel) must decode like any other."""
    import torch
    frames, subs, res, total = synthetic.flac_batch(91, 96, block_size=576, orders=(12, 7))
    want = oraclelib.flac_transform(frames, subs, res, total)
    d_fr = torch.from_numpy(frames.view(np.uint8).copy()).to(gpu)
    d_sf = torch.from_numpy(subs.view(np.uint8).copy()).to(gpu)
    d_res = torch.from_numpy(res).to(gpu)
    h_fr = torch.zeros(d_fr.numel(), dtype=torch.uint8).pin_memory()          # host-visible to the device, all zero for now
    h_sf = torch.zeros(d_sf.numel(), dtype=torch.uint8).pin_memory()
    out = torch.full((total,), -7, dtype=torch.int32, device=gpu)
    torch.cuda.synchronize()
    torch.cuda._sleep(400_000_000)                                              # ~0.2 s of device time in front of the copies
    h_fr.copy_(d_fr, non_blocking=True)
    h_sf.copy_(d_sf, non_blocking=True)
    assert not h_fr.any()                                                       # the host still sees the stale records
    rc = afgpu.lib().afg_flac_transform_hip(len(frames), h_fr.data_ptr(), h_sf.data_ptr(), d_res.data_ptr(), out.data_ptr(), None,
                                            torch.cuda.current_stream().cuda_stream)
    assert rc == 0, afgpu.lib().afg_last_error()
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)


def _encode_against(samples, coef, shift):
    """residuals that make the reference's 32-bit recurrence (drflac.d:1060-1099, :1235) reproduce `samples`"""
    order = len(coef)
    res = samples.astype(np.int64).copy()
    for t in range(order, len(samples)):
        acc = int(np.dot(coef.astype(np.int64), samples[t - order:t][::-1].astype(np.int64)))
        acc = (acc + 2 ** 31) % 2 ** 32 - 2 ** 31                    # the int32 accumulator wraps
        res[t] = (int(samples[t]) - (acc >> shift) + 2 ** 31) % 2 ** 32 - 2 ** 31
    return res.astype(np.int32)


@pytest.mark.parametrize("order", [2, 7, 12, 31])
def test_flac_in_range_samples_with_a_wrapping_sum(gpu, order):
    """Valid-looking 16-bit channels whose prediction sum passes 2^31: coefficients at the int16 limits, every decoded sample
    inside int16 (the residuals are computed against the reference's wrapping int32 recurrence).  Frames 40..47 hold samples
    far beyond int16 as well.  (Written for a packed int16 dot-product form of the recurrence, which turned out slower
    than the 32-bit multiplies -- HISTORY.md section 8 -- and is not in the product; the cases stay.)"""
    rng = np.random.default_rng(order)
    n_frames, bs = 130, 192
    frames = np.zeros(n_frames, afgpu.FLAC_FRAME_DTYPE)
    subs = np.zeros(2 * n_frames, afgpu.FLAC_SUBFRAME_DTYPE)
    res = np.zeros((n_frames, 2, bs), np.int32)
    for f in range(n_frames):
        frames[f] = (f * 2 * bs, f * 2 * bs, bs, 2 * f, 2, afgpu.FLAC_INDEPENDENT, 16, 0, [0] * 4)
        for c in range(2):
            coef = rng.choice(np.array([32767, -32768, 30000, -29000, 12345]), order).astype(np.int16)
            shift = int(rng.integers(0, 16))
            wild = 40 <= f < 48 and c == 0
            smp = rng.integers(-32768, 32768, bs) if not wild else rng.integers(-2 ** 20, 2 ** 20, bs)
            sf = subs[2 * f + c]
            sf["coef"][:order] = coef
            sf["order"], sf["shift"], sf["wasted"], sf["use64"] = order, shift, 0, 0
            res[f, c] = _encode_against(smp, coef, shift)
    flat = res.reshape(-1)
    total = n_frames * 2 * bs
    want = oraclelib.flac_transform(frames, subs, flat, total)
    # the construction itself: the oracle reproduces the chosen samples (left-justified by 32 - bps)
    got_i32, _ = run_gpu(gpu, frames, subs, flat, total, want_float=False)
    assert np.array_equal(got_i32, want)
    assert np.abs(flat.astype(np.int64)).max() > 2 ** 28                # sums that wrapped: the residuals had to undo them


@pytest.mark.parametrize("ch", [3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("kw", [
    dict(n_frames=75, block_size=4096, orders=(8, 12)),                                  # several passes per block of 32 frames, a ragged last block
    dict(n_frames=40, vary_block=True, orders=(2, 8, 12, 31)),                            # block sizes differ inside a pass, odd sizes, order 31
    dict(n_frames=33, block_size=1152, bps=24, orders=(12, 32), residual_scale=3000.0),   # wide sums, order 32
])
def test_flac_multichannel_kernel_bit_exact(gpu, ch, kw):
    """groups of frames with one channel count above two: flac_restore_mc_kernel (whole interleaved frames per step), int32
    rows and int16 rows, both outputs"""
    frames, subframes, res, total = synthetic.flac_batch(100 + ch, channels=ch, **kw)
    want_i, want_f = oraclelib.flac_transform(frames, subframes, res, total, want_float=True)
    got_i, got_f = run_gpu(gpu, frames, subframes, res, total)
    assert (got_i == want_i).all(), f"{int((got_i != want_i).sum())} int32 mismatches"
    assert (got_f.view(np.uint32) == want_f.view(np.uint32)).all()
    mask = afgpu.flac_variants(frames, subframes)
    assert mask & 0xf00 and not (mask & 0xff), hex(mask)           # only instantiations of the multi-channel kernel
    if kw.get("bps", 16) == 16:
        res16 = np.clip(res, -30000, 30000).astype(np.int32)
        fr16, packed = synthetic.flac_pack16(frames, res16)
        if (fr16["res16"] != 0).all():
            want16 = oraclelib.flac_transform(fr16, subframes, packed, total)
            got16, _ = run_gpu(gpu, fr16, subframes, packed, total, want_float=False)
            assert (got16 == want16).all()


def test_flac_multichannel_and_stereo_groups_in_one_call(gpu):
    """a batch whose 32-frame groups are of different kinds -- stereo, six channels, a group that mixes channel counts (the
    general step's), mono -- goes to the right kernels group by group, with and without the variants mask"""
    import torch
    parts = [synthetic.flac_batch(7, n_frames=64, block_size=1024, channels=2, orders=(8, 12)),
             synthetic.flac_batch(8, n_frames=64, block_size=1024, channels=6, orders=(8, 12)),
             synthetic.flac_batch(9, n_frames=16, block_size=1024, channels=6, orders=(8, 12)),
             synthetic.flac_batch(10, n_frames=16, block_size=1024, channels=3, orders=(8, 12)),      # with the 16 before: a mixed group
             synthetic.flac_batch(11, n_frames=40, block_size=1024, channels=1, orders=(8, 12)),
             synthetic.flac_batch(12, n_frames=32, block_size=1024, channels=4, orders=(2, 32))]
    frames, subs, ress = [], [], []
    in_off = out_off = sf_off = 0
    for fr, sb, rs, tot in parts:
        fr = fr.copy()
        fr["in_off"] += np.uint64(in_off); fr["out_off"] += np.uint64(out_off); fr["sf_index"] += np.uint32(sf_off)
        frames.append(fr); subs.append(sb); ress.append(rs)
        in_off += len(rs); out_off += tot; sf_off += len(sb)
    frames, subs, res = np.concatenate(frames), np.concatenate(subs), np.concatenate(ress)
    want = oraclelib.flac_transform(frames, subs, res, out_off)
    mask = afgpu.flac_variants(frames, subs)
    assert mask & 0xf00 and mask & 0xff
    d_fr = torch.from_numpy(frames.view(np.uint8).copy()).to(gpu)
    d_sf = torch.from_numpy(subs.view(np.uint8).copy()).to(gpu)
    d_res = torch.from_numpy(res).to(gpu)
    for variants in (None, mask):
        out = torch.full((out_off,), -777, dtype=torch.int32, device=gpu)
        afgpu.flac_transform(len(frames), d_fr, d_sf, d_res, out, None, None, variants=variants)
        torch.cuda.synchronize()
        assert (out.cpu().numpy() == want).all()
