"""QOA encoder (output side): the HIP path's file bytes against the CPU oracle's restatement of qoa_encode_frame /
QOAEncoder (qoa.d:295-399, :538-700), through the C ABI.  Integer work: byte-exact."""
import numpy as np
import pytest

import afgpu
import oraclelib

pytestmark = pytest.mark.gpu


def make_pcm(rng, frames, channels):
    t = np.arange(frames)[:, None]
    x = 9000 * np.sin(0.01 * (1 + np.arange(channels))[None, :] * t) + 3000 * rng.standard_normal((frames, channels))
    x[frames // 3: frames // 3 + 50] *= 4.0                    # a loud burst: clamping and the large scalefactors
    return np.clip(x, -32768, 32767).astype(np.int16)


def encode_gpu(gpu, blocks, rate, as_float=False):
    import torch
    recs, n_in, n_out = afgpu.qoa_encode_layout([b.shape for b in blocks], rate)
    flat = np.concatenate([b.reshape(-1) for b in blocks]) if blocks else np.zeros(0, np.int16)
    d_out = torch.zeros(max(n_out, 8), dtype=torch.uint8, device=gpu)
    d_recs = torch.from_numpy(recs.view(np.uint8).copy()).to(gpu)
    if as_float:
        d_in = torch.from_numpy(flat.astype(np.float32)).to(gpu)
        afgpu.qoa_encode(len(recs), d_recs, d_out, d_pcm_f32=d_in)
    else:
        d_in = torch.from_numpy(flat.copy()).to(gpu)
        afgpu.qoa_encode(len(recs), d_recs, d_out, d_pcm_i16=d_in)
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    return [out[int(r["out_off"]): int(r["out_off"]) + afgpu.qoa_encoded_size(int(r["samples"]), int(r["channels"]))] for r in recs]


@pytest.mark.parametrize("shapes", [
    [(5120 * 2 + 777, 2), (100, 1), (5120, 1)],              # ragged last frame and slice, a sub-frame stream, exact frame
    [(3000, 3), (6001, 8), (41, 5), (20, 4)],                # channel batches of 4 (+ idle lanes), 5..8 channels
    [(1, 1), (19, 2), (21, 2)],                              # shorter than a slice
])
def test_qoa_encode_matches_oracle(gpu, shapes):
    rng = np.random.default_rng(11)
    blocks = [make_pcm(rng, n, c) for n, c in shapes]
    got = encode_gpu(gpu, blocks, 44100)
    for b, g in zip(blocks, got):
        want, _ = oraclelib.qoa_encode(b, 44100)
        assert len(g) == len(want)
        assert np.array_equal(g, want), int((g != want).sum())


def test_qoa_encode_float_input_is_the_writer_conversion(gpu):
    rng = np.random.default_rng(5)
    x = np.clip(rng.standard_normal((5120 + 300, 2)) * 0.3, -1, 1).astype(np.float32)
    import torch
    recs, n_in, n_out = afgpu.qoa_encode_layout([x.shape], 48000)
    d_out = torch.zeros(n_out, dtype=torch.uint8, device=gpu)
    afgpu.qoa_encode(1, torch.from_numpy(recs.view(np.uint8).copy()).to(gpu), d_out, d_pcm_f32=torch.from_numpy(x.reshape(-1)).to(gpu))
    torch.cuda.synchronize()
    s = (32768.5 + x.astype(np.float64) * 32767.0).astype(np.int64) - 32768                # qoa.d:632-636
    want, _ = oraclelib.qoa_encode(s.astype(np.int16), 48000)
    assert np.array_equal(d_out.cpu().numpy()[:len(want)], want)


def test_qoa_encode_then_decode_round_trip(gpu):
    """encode (HIP) -> open as a stream (HIP decode) returns the encoder's own reconstruction."""
    rng = np.random.default_rng(9)
    pcm = make_pcm(rng, 5120 * 3 + 123, 2)
    data = encode_gpu(gpu, [pcm], 32000)[0].tobytes()
    _, recon = oraclelib.qoa_encode(pcm, 32000)
    st = afgpu.AudioStream()
    st.openFromMemory(data)
    assert not st.isError(), st.errorMessage()
    assert st.getFormat() == afgpu.FORMAT_QOA
    assert st.getNumChannels() == 2 and st.getSamplerate() == 32000 and st.getLengthInFrames() == len(pcm)
    got = np.zeros(pcm.size, np.float32)
    assert st.readSamplesFloat(got) == len(pcm)
    assert np.array_equal(got.reshape(-1, 2), recon.astype(np.float32) * np.float32(1.0 / 32767.0))


def test_qoa_encode_rejects_bad_calls(gpu):
    import torch
    d = torch.zeros(64, dtype=torch.uint8, device=gpu)
    with pytest.raises(afgpu.AfgError):
        afgpu.qoa_encode(1, d, d)                                            # neither input
    with pytest.raises(afgpu.AfgError):
        afgpu.qoa_encode(1, d, d, d_pcm_i16=d, d_pcm_f32=d)                  # both inputs
