"""CELT transform stage in tolerance mode (AFG_NUMERIC_TOLERANCE, csrc/celt_walk.hip): the segment walk with the
de-emphasis recurrence (dopus.d:3695-3701) re-associated into a prefix sum and streams cut where the post-filter is
provably idle (dopus.d:3294-3296, :3333).  Checked against the CPU oracle within north_star's tolerance -- 1e-5 RMS on
the API scale -- and, after OpusFile.readFrame's int16 conversion (dopus.d:7923-7926), by the rate of samples that land on
a neighbouring int16 value (Opus is judged on that rate)."""
import numpy as np
import pytest

import afgpu
import oraclelib
from afgpu import synthetic

pytestmark = [pytest.mark.gpu, pytest.mark.numeric_tolerance]

TOL_RMS = 1e-5          # north_star: float output within 1e-5 RMS of the reference decoders


@pytest.fixture(autouse=True)
def no_path_override(monkeypatch):
    monkeypatch.delenv("AFG_CELT_PATH", raising=False)          # the numeric mode (default: tolerance) picks the walk
    # (with thousands of channel pairs the library walks sequences of up to 512 frames whole; these batches are small, the
    # setting only pins the policy the segmentation tests are written against)
    monkeypatch.setenv("AFG_CELT_WHOLE_FRAMES", "0")


def run_gpu(gpu, rec_base, recs, coeffs, total, states=None):
    import torch
    d_out = torch.full((total,), float("nan"), dtype=torch.float32, device=gpu)
    d_states = None if states is None else torch.from_numpy(states).to(gpu)
    afgpu.celt_transform(len(rec_base) - 1, torch.from_numpy(rec_base.view(np.int64)).to(gpu),
                         torch.from_numpy(recs.view(np.uint8).copy()).to(gpu), torch.from_numpy(coeffs).to(gpu),
                         d_out, d_states)
    torch.cuda.synchronize()
    return d_out.cpu().numpy(), (None if states is None else d_states.cpu().numpy())


def check(got, want, scale_rms=True):
    """RMS within tolerance; returns (rms, relative rms, int16 flip rate, largest int16 step)."""
    assert not np.isnan(got).any(), "unwritten output"
    diff = got.astype(np.float64) - want
    rms = float(np.sqrt(np.mean(diff ** 2)))
    ref = float(np.sqrt(np.mean(want.astype(np.float64) ** 2))) or 1.0
    assert rms <= TOL_RMS, (rms, ref)
    gi, _ = oraclelib.opus_output(got)
    wi, _ = oraclelib.opus_output(want)
    step = np.abs(gi.astype(np.int32) - wi.astype(np.int32))
    return rms, rms / ref, float((step != 0).mean()), int(step.max())


CASES = [
    dict(frames_per_stream=[6, 3, 9], channels=[2, 1, 2]),
    dict(frames_per_stream=[12], channels=[2], p_transient=1.0, p_postfilter=1.0),
    dict(frames_per_stream=[10, 7], channels=[1, 2], frame_sizes=(120, 240, 480, 960), p_transient=0.5, p_postfilter=0.6),
    dict(frames_per_stream=[8], channels=[1], p_postfilter=0.0),
    dict(frames_per_stream=[3] * 42, channels=[2] * 37 + [1] * 5, p_postfilter=0.5),
]


@pytest.mark.parametrize("seg", [0, 8, 128])
@pytest.mark.parametrize("kw", CASES)
def test_walk_matches_oracle_within_tolerance(gpu, monkeypatch, kw, seg):
    monkeypatch.setenv("AFG_CELT_SEG_RECS", str(seg))
    rec_base, recs, coeffs, total = synthetic.celt_batch(5, **kw)
    coeffs = (coeffs * 0.05).astype(np.float32)                  # programme level: rms about 0.1 of full scale
    want = oraclelib.celt_transform(rec_base, recs, coeffs, total)
    got, _ = run_gpu(gpu, rec_base, recs, coeffs, total)
    rms, rel, flips, step = check(got, want)
    assert rel < 1e-6 and flips < 0.01 and step <= 1, (rms, rel, flips, step)


@pytest.mark.parametrize("seg", [4, 16, 64, 128])
@pytest.mark.parametrize("kw", [
    dict(frames_per_stream=[300, 41, 170], channels=[2, 2, 1], p_postfilter=0.2),
    dict(frames_per_stream=[257, 90], channels=[2, 1], p_postfilter=0.1, p_transient=0.4),
    dict(frames_per_stream=[400], channels=[2], frame_sizes=(120, 240, 480, 960), p_postfilter=0.08),
    dict(frames_per_stream=[120, 120], channels=[1, 1], p_postfilter=0.15),              # two mono streams sharing a wavefront pair slot
])
def test_walk_segments_are_exact_cuts(gpu, monkeypatch, kw, seg):
    """Long streams cut into segments: a segment starts from nothing a few frames before a cut (the warm-up) and must
    reproduce what the walk from the stream's first frame gives -- the cut conditions make the post-filter exact, the
    de-emphasis memory is rebuilt to 0.85^1026.  Compared with the oracle AND, bit for bit, with the one-segment walk."""
    rec_base, recs, coeffs, total = synthetic.celt_batch(77 + seg, **kw)
    coeffs = (coeffs * 0.05).astype(np.float32)
    want = oraclelib.celt_transform(rec_base, recs, coeffs, total)
    monkeypatch.setenv("AFG_CELT_SEG_RECS", "0")
    whole, _ = run_gpu(gpu, rec_base, recs, coeffs, total)
    monkeypatch.setenv("AFG_CELT_SEG_RECS", str(seg))
    got, _ = run_gpu(gpu, rec_base, recs, coeffs, total)
    rms, rel, flips, step = check(got, want)
    assert rel < 1e-6 and flips < 0.01 and step <= 1, (rms, rel, flips, step)
    # the cuts are exact and the warm-up leaves nothing of its start in a float: the same bits as the unsegmented walk
    assert np.array_equal(got.view(np.uint32), whole.view(np.uint32)), int((got.view(np.uint32) != whole.view(np.uint32)).sum())


def test_walk_short_period_postfilter(gpu):
    rec_base, recs, coeffs, total = synthetic.celt_batch(8, [6], [2], p_postfilter=1.0)
    coeffs = (coeffs * 0.05).astype(np.float32)
    recs["pf_period_new"] = np.where(np.arange(len(recs)) % 2, 15, 16)       # minimum lags: 13-sample steps
    want = oraclelib.celt_transform(rec_base, recs, coeffs, total)
    got, _ = run_gpu(gpu, rec_base, recs, coeffs, total)
    check(got, want)


@pytest.mark.parametrize("seg", [0, 6])
def test_walk_chunked_with_state_equals_whole(gpu, monkeypatch, seg):
    """(With a carry state every channel pair is walked by one wavefront whatever the item size asked for: the state
    is read at the first frame and rewritten in place after the last.)"""
    monkeypatch.setenv("AFG_CELT_SEG_RECS", str(seg))
    rec_base, recs, coeffs, total = synthetic.celt_batch(9, [40], [2], p_postfilter=0.3, p_transient=0.3)
    coeffs = (coeffs * 0.05).astype(np.float32)
    want = oraclelib.celt_transform(rec_base, recs, coeffs, total)
    states = np.zeros((2, afgpu.CELT_STATE_FLOATS), np.float32)
    out = np.full(total, np.nan, np.float32)
    for lo, hi in ((0, 4), (4, 23), (23, 40)):
        sel = np.concatenate([np.arange(int(rec_base[c]) + lo, int(rec_base[c]) + hi) for c in range(2)])
        rb = np.array([0, hi - lo, 2 * (hi - lo)], np.uint64)
        got, st = run_gpu(gpu, rb, recs[sel].copy(), coeffs, total, states.reshape(-1).copy())
        states = st.reshape(2, -1)
        m = ~np.isnan(got)
        out[m] = got[m]
    check(out, want)


def test_walk_state_blob_is_interchangeable_with_the_exact_paths(gpu, monkeypatch):
    """A stream may change numeric mode between chunks: the carry state has one layout."""
    rec_base, recs, coeffs, total = synthetic.celt_batch(19, [12], [2], p_postfilter=0.7)
    coeffs = (coeffs * 0.05).astype(np.float32)
    want = oraclelib.celt_transform(rec_base, recs, coeffs, total)
    states = np.zeros((2, afgpu.CELT_STATE_FLOATS), np.float32)
    out = np.full(total, np.nan, np.float32)
    for path, (lo, hi) in zip(("stream", "walk", "split", "walk"), ((0, 3), (3, 6), (6, 9), (9, 12))):
        monkeypatch.setenv("AFG_CELT_PATH", path)                   # stream / split: the bit-exact kernels
        sel = np.concatenate([np.arange(int(rec_base[c]) + lo, int(rec_base[c]) + hi) for c in range(2)])
        rb = np.array([0, hi - lo, 2 * (hi - lo)], np.uint64)
        got, st = run_gpu(gpu, rb, recs[sel].copy(), coeffs, total, states.reshape(-1).copy())
        states = st.reshape(2, -1)
        m = ~np.isnan(got)
        out[m] = got[m]
    check(out, want)


def test_walk_padded_output_stride(gpu):
    """out_stride 3 (stereo written into a 3-channel plane) and an unaligned mono row: per-lane path."""
    rec_base, recs, coeffs, total = synthetic.celt_batch(22, [4, 4], [2, 1])
    coeffs = (coeffs * 0.05).astype(np.float32)
    stereo = recs["out_stride"] == 2
    recs["out_off"][stereo] = (recs["out_off"][stereo] // 2) * 3 + recs["out_off"][stereo] % 2
    recs["out_stride"][stereo] = 3
    base = int(recs["out_off"][stereo].max()) + 3 * 960
    mono = ~stereo
    recs["out_off"][mono] = recs["out_off"][mono] - recs["out_off"][mono].min() + base + 1       # odd offset
    total = int(recs["out_off"][mono].max()) + 960
    want = oraclelib.celt_transform(rec_base, recs, coeffs, total)
    got, _ = run_gpu(gpu, rec_base, recs, coeffs, total)
    written = np.zeros(total, bool)
    for r in recs:
        written[int(r["out_off"]) + np.arange(int(r["frame_size"])) * int(r["out_stride"])] = True
    assert not np.isnan(got[written]).any() and np.isnan(got[~written]).all()
    check(got[written], want[written])


def test_walk_stereo_pair_two_mod_four_into_the_plane(gpu):
    """A stereo stream whose frames start 2 mod 4 floats into the output plane (an odd number of stereo frames in front of
    it): 8-byte aligned only, so the hot path's 16-byte stores do not apply."""
    rec_base, recs, coeffs, total = synthetic.celt_batch(23, [5, 4], [2, 2], frame_sizes=(960,))
    coeffs = (coeffs * 0.05).astype(np.float32)
    recs = recs.copy()
    recs["out_off"] += np.uint64(2)                                # every frame of both streams: offset = 2 (mod 4)
    total += 2
    assert (recs["out_off"][recs["out_stride"] == 2] % 4 >= 2).all()
    want = oraclelib.celt_transform(rec_base, recs, coeffs, total)
    got, _ = run_gpu(gpu, rec_base, recs, coeffs, total)
    assert np.isnan(got[:2]).all()
    check(got[2:], want[2:])


def test_walk_at_full_scale_amplitude(gpu):
    """The generators' native level is 5 dB over full scale (rms 1.8): the absolute tolerance still holds there."""
    rec_base, recs, coeffs, total = synthetic.celt_batch(31, [60, 25], [2, 2], p_postfilter=0.5, p_transient=0.2)
    want = oraclelib.celt_transform(rec_base, recs, coeffs, total)
    got, _ = run_gpu(gpu, rec_base, recs, coeffs, total)
    rms, rel, flips, step = check(got, want)
    assert rel < 1e-6, (rms, rel)


def test_numeric_mode_switch(gpu, monkeypatch):
    """afg_set_numeric_mode selects the path when AFG_CELT_PATH does not: exact -> bit-identical to the oracle."""
    rec_base, recs, coeffs, total = synthetic.celt_batch(3, [9, 5], [2, 1], p_postfilter=0.5)
    want = oraclelib.celt_transform(rec_base, recs, coeffs, total)
    prev = afgpu.set_numeric_mode(afgpu.NUMERIC_EXACT)
    try:
        assert afgpu.get_numeric_mode() == afgpu.NUMERIC_EXACT
        got, _ = run_gpu(gpu, rec_base, recs, coeffs, total)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        afgpu.set_numeric_mode(afgpu.NUMERIC_TOLERANCE)
        got, _ = run_gpu(gpu, rec_base, recs, coeffs, total)
        check(got, want)
        assert prev == afgpu.NUMERIC_TOLERANCE                      # the default
    finally:
        afgpu.set_numeric_mode(afgpu.NUMERIC_FROM_ENV)


@pytest.mark.parametrize("seed", range(6))
def test_walk_random_batches_and_item_sizes(gpu, monkeypatch, seed):
    """Random batch shapes -- empty sequences, one-frame streams, mono / stereo mixes, an odd channel at the end, frame-size
    mixes -- under random item sizes (including ones that are not a multiple of anything): every segmentation must give the
    bits of the unsegmented walk, which must be within tolerance of the oracle."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 40))
    fps = [int(x) for x in rng.choice([0, 1, 2, 3, 5, 9, 17, 40, 90], n)]
    if sum(fps) == 0:
        fps[0] = 7
    chans = [int(x) for x in rng.choice([1, 2, 2], n)]
    sizes = (960,) if seed % 2 == 0 else (120, 240, 480, 960)
    rec_base, recs, coeffs, total = synthetic.celt_batch(500 + seed, fps, chans, frame_sizes=sizes,
                                                         p_postfilter=float(rng.choice([0.0, 0.1, 0.3])), p_transient=0.2)
    coeffs = (coeffs * 0.05).astype(np.float32)
    want = oraclelib.celt_transform(rec_base, recs, coeffs, total)
    monkeypatch.setenv("AFG_CELT_SEG_RECS", "0")
    whole, _ = run_gpu(gpu, rec_base, recs, coeffs, total)
    rms, rel, flips, step = check(whole, want)
    assert rel < 1e-6 and flips < 0.01 and step <= 1, (rms, rel, flips, step)
    for seg in [int(x) for x in rng.choice([1, 2, 3, 5, 7, 12, 31, 64, 100, 1000], 4, replace=False)]:
        monkeypatch.setenv("AFG_CELT_SEG_RECS", str(seg))
        got, _ = run_gpu(gpu, rec_base, recs, coeffs, total)
        assert np.array_equal(got.view(np.uint32), whole.view(np.uint32)), (seed, seg, int((got.view(np.uint32) != whole.view(np.uint32)).sum()))


def test_walk_many_short_streams(gpu):
    """20 000 channel sequences of 1-3 frames: the search in rec_base, items that span hundreds of channel pairs."""
    rng = np.random.default_rng(77)
    n = 10000
    fps = [int(x) for x in rng.integers(1, 4, n)]
    rec_base, recs, coeffs, total = synthetic.celt_batch(909, fps, [2] * n, p_postfilter=0.3)
    coeffs = (coeffs * 0.05).astype(np.float32)
    want = oraclelib.celt_transform(rec_base, recs, coeffs, total)
    got, _ = run_gpu(gpu, rec_base, recs, coeffs, total)
    rms, rel, flips, step = check(got, want)
    assert rel < 1e-6 and flips < 0.01 and step <= 1, (rms, rel, flips, step)


def test_walk_default_policy_short_whole_long_cut(gpu, monkeypatch):
    """Sequences of up to AFG_CELT_WHOLE_FRAMES frames walked whole, longer ones in segments (what the library does by itself
    from 8192 channel pairs up) -- same bits as the unsegmented walk either way."""
    rec_base, recs, coeffs, total = synthetic.celt_batch(4242, [700, 90, 513, 512, 30], [2, 2, 1, 2, 1], p_postfilter=0.2)
    coeffs = (coeffs * 0.05).astype(np.float32)
    want = oraclelib.celt_transform(rec_base, recs, coeffs, total)
    monkeypatch.setenv("AFG_CELT_SEG_RECS", "0")
    whole, _ = run_gpu(gpu, rec_base, recs, coeffs, total)
    monkeypatch.delenv("AFG_CELT_SEG_RECS")
    monkeypatch.setenv("AFG_CELT_WHOLE_FRAMES", "512")
    got, _ = run_gpu(gpu, rec_base, recs, coeffs, total)
    check(got, want)
    assert np.array_equal(got.view(np.uint32), whole.view(np.uint32))
