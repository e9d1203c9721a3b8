"""CELT transform stage: HIP path vs the CPU oracle through the C ABI (north-star tolerance 1e-5 RMS
on the API scale; the expression trees are the reference's, so the match is expected to be bitwise)."""
import numpy as np
import pytest

import afgpu
import oraclelib
from afgpu import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["split", "stream"])
def celt_path(request, monkeypatch):
    """Both device paths on every case: 'split' (record-parallel iMDCT kernel + per-stream post-filter kernel, the
    choice for few streams) and 'stream' (one fused walk per stream, the choice for device-filling batches)."""
    monkeypatch.setenv("AFG_CELT_PATH", request.param)
    return request.param


def run_gpu(gpu, rec_base, recs, coeffs, total, states=None):
    import torch
    d_out = torch.full((total,), float("nan"), dtype=torch.float32, device=gpu)
    d_states = None if states is None else torch.from_numpy(states).to(gpu)
    afgpu.celt_transform(len(rec_base) - 1, torch.from_numpy(rec_base.view(np.int64)).to(gpu),
                         torch.from_numpy(recs.view(np.uint8).copy()).to(gpu), torch.from_numpy(coeffs).to(gpu),
                         d_out, d_states)
    torch.cuda.synchronize()
    return d_out.cpu().numpy(), (None if states is None else d_states.cpu().numpy())


def check(got, want):
    assert not np.isnan(got).any(), "unwritten output"
    rms = float(np.sqrt(np.mean((got.astype(np.float64) - want) ** 2)))
    assert rms <= 1e-5, rms
    return int((got.view(np.uint32) != want.view(np.uint32)).sum())


@pytest.mark.parametrize("kw", [
    dict(frames_per_stream=[6, 3, 9], channels=[2, 1, 2]),
    dict(frames_per_stream=[12], channels=[2], p_transient=1.0, p_postfilter=1.0),
    dict(frames_per_stream=[10, 7], channels=[1, 2], frame_sizes=(120, 240, 480, 960), p_transient=0.5, p_postfilter=0.6),
    dict(frames_per_stream=[8], channels=[1], p_postfilter=0.0),
])
def test_celt_matches_oracle(gpu, kw):
    rec_base, recs, coeffs, total = synthetic.celt_batch(5, **kw)
    want = oraclelib.celt_transform(rec_base, recs, coeffs, total)
    got, _ = run_gpu(gpu, rec_base, recs, coeffs, total)
    assert check(got, want) == 0


def test_celt_short_period_postfilter(gpu):
    rec_base, recs, coeffs, total = synthetic.celt_batch(8, [6], [2], p_postfilter=1.0)
    recs["pf_period_new"] = np.where(np.arange(len(recs)) % 2, 15, 16)       # minimum lags: 13-sample steps
    want = oraclelib.celt_transform(rec_base, recs, coeffs, total)
    got, _ = run_gpu(gpu, rec_base, recs, coeffs, total)
    assert check(got, want) == 0


def test_celt_chunked_with_state_equals_whole(gpu):
    rec_base, recs, coeffs, total = synthetic.celt_batch(9, [10], [2], p_postfilter=0.7, p_transient=0.3)
    want = oraclelib.celt_transform(rec_base, recs, coeffs, total)
    states = np.zeros((2, afgpu.CELT_STATE_FLOATS), np.float32)
    out = np.full(total, np.nan, np.float32)
    for lo, hi in ((0, 4), (4, 10)):
        sel = np.concatenate([np.arange(int(rec_base[c]) + lo, int(rec_base[c]) + hi) for c in range(2)])
        rb = np.array([0, hi - lo, 2 * (hi - lo)], np.uint64)
        got, st = run_gpu(gpu, rb, recs[sel].copy(), coeffs, total, states.reshape(-1).copy())
        states = st.reshape(2, -1)
        m = ~np.isnan(got)
        out[m] = got[m]
    assert check(out, want) == 0


def test_celt_many_streams_row_path(gpu):
    """37 stereo + 5 mono streams of equal frame size: the de-emphasis pass walks whole rows (several
    wavefronts, a partly filled last one, a mono/stereo boundary inside a wavefront)."""
    rec_base, recs, coeffs, total = synthetic.celt_batch(21, [3] * 42, [2] * 37 + [1] * 5, p_postfilter=0.5)
    want = oraclelib.celt_transform(rec_base, recs, coeffs, total)
    got, _ = run_gpu(gpu, rec_base, recs, coeffs, total)
    assert check(got, want) == 0


@pytest.mark.parametrize("seq,duo", [(2, 0), (4, 0), (8, 0), (16, 0), (32, 0), (2, 1), (4, 1), (8, 1)])
def test_celt_deemph_sequences_per_wavefront(gpu, monkeypatch, seq, duo):
    """Every instantiation of the de-emphasis pass (2 ... 32 chain lanes per wavefront, as one wavefront or as a mover /
    chainer pair; the host picks one from the number of sequences) on streams of different lengths -- lanes whose sequence
    has ended sit the later steps out -- with mixed frame sizes in one stream and a mono / stereo mix."""
    monkeypatch.setenv("AFG_CELT_DE_SEQ", str(seq))
    monkeypatch.setenv("AFG_CELT_DE_DUO", str(duo))
    fps = [3, 9, 5, 9, 2, 7, 4, 11, 6, 3, 8, 5, 10, 4, 7, 6, 9, 3, 5]
    rec_base, recs, coeffs, total = synthetic.celt_batch(31 + seq, fps, [2] * 15 + [1] * 4, p_postfilter=0.5, p_transient=0.3)
    want = oraclelib.celt_transform(rec_base, recs, coeffs, total)
    got, _ = run_gpu(gpu, rec_base, recs, coeffs, total)
    assert check(got, want) == 0
    rec_base, recs, coeffs, total = synthetic.celt_batch(41 + seq, [6, 4, 9], [2, 2, 2], frame_sizes=(120, 240, 480, 960), p_postfilter=0.5)
    want = oraclelib.celt_transform(rec_base, recs, coeffs, total)
    got, _ = run_gpu(gpu, rec_base, recs, coeffs, total)
    assert check(got, want) == 0


def test_celt_padded_output_stride(gpu):
    """out_stride 3 (stereo written into a 3-channel plane) and an unaligned mono row: per-lane path."""
    rec_base, recs, coeffs, total = synthetic.celt_batch(22, [4, 4], [2, 1])
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
    assert np.array_equal(got[written].view(np.uint32), want[written].view(np.uint32))


def opus_output_inputs():
    rng = np.random.default_rng(17)
    k = np.arange(-40000, 40000, 7, dtype=np.float64)
    ties = ((k + 0.5) / 32768.0).astype(np.float32)                      # exact .5 cases: ties go to even
    return np.concatenate([ties, (rng.standard_normal(50000) * 0.4).astype(np.float32),
                           np.array([0, -0.0, 1.0, -1.0, 0.99998474, 1.5, -1.5, 3e4, -3e4, 1e-9, 32767 / 32768], np.float32)])


def test_opus_output_conversion_matches_oracle(gpu):
    """OpusFile.readFrame's Float2IntScaled + saturation and stream.d:480, element-wise, bit-exact."""
    import torch
    x = opus_output_inputs()
    want_i, want_f = oraclelib.opus_output(x)
    d_in = torch.from_numpy(x).to(gpu)
    d_i = torch.zeros(x.size, dtype=torch.int16, device=gpu)
    d_f = torch.zeros(x.size, dtype=torch.float32, device=gpu)
    afgpu.opus_output(x.size, d_in, d_i, d_f)
    afgpu.opus_output(x.size, d_in, None, d_in)                           # in place, float only
    torch.cuda.synchronize()
    assert np.array_equal(d_i.cpu().numpy(), want_i)
    assert np.array_equal(d_f.cpu().numpy().view(np.uint32), want_f.view(np.uint32))
    assert np.array_equal(d_in.cpu().numpy().view(np.uint32), want_f.view(np.uint32))
