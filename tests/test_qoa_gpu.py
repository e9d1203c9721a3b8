"""QOA frame decode: HIP path vs the CPU oracle (bit-exact int16 and float), through the C ABI."""
import numpy as np
import pytest

import afgpu
import oraclelib
from test_oracle_qoa import tone

pytestmark = pytest.mark.gpu


def batch(files):
    """Concatenate QOA files into one byte plane + frame table."""
    recs, planes, byte_base, out_base = [], [], 0, 0
    for pcm in files:
        data, _ = oraclelib.qoa_encode(pcm)
        fr, ch, _, total = afgpu.qoa_frames(data.tobytes(), out_base, byte_base)
        recs.append(fr)
        planes.append(data)
        pad = (-data.size) % 8
        if pad:
            planes.append(np.zeros(pad, np.uint8))
        byte_base += data.size + pad
        out_base += total * ch
    return np.concatenate(recs), np.concatenate(planes), out_base


@pytest.mark.parametrize("shapes", [[(5120 * 3 + 100, 2)], [(700, 1), (5120 * 2, 2), (19, 2), (9000, 1)],
                                    [(6000, 3), (5200, 5), (4000, 8)], [(5120 * 40, 2)]])
def test_qoa_bit_exact(gpu, shapes):
    import torch
    files = [tone(n, ch, seed=i * 7 + n) for i, (n, ch) in enumerate(shapes)]
    frames, data, total = batch(files)
    want_i, want_f = oraclelib.qoa_transform(frames, data, total)
    d_i = torch.full((total,), -7, dtype=torch.int16, device=gpu)
    d_f = torch.full((total,), float("nan"), dtype=torch.float32, device=gpu)
    afgpu.qoa_transform(len(frames), torch.from_numpy(frames.view(np.uint8).copy()).to(gpu),
                        torch.from_numpy(data).to(gpu), d_i, d_f)
    torch.cuda.synchronize()
    assert (d_i.cpu().numpy() == want_i).all()
    assert (d_f.cpu().numpy().view(np.uint32) == want_f.view(np.uint32)).all()


def test_qoa_wrapping_lms_state(gpu):
    """Random slice words and extreme LMS state: the int arithmetic must wrap like the reference."""
    import torch
    rng = np.random.default_rng(3)
    files = [tone(5120 * 2, 2, seed=1)]
    frames, data, total = batch(files)
    data = data.copy()
    for fr in frames:
        b = int(fr["byte_off"])
        data[b + 8:b + 8 + 32] = rng.integers(0, 256, 32, dtype=np.uint8)               # LMS state
        n = 8 * 256 * 2
        data[b + 40:b + 40 + n] = rng.integers(0, 256, n, dtype=np.uint8)               # slices
    want_i = oraclelib.qoa_transform(frames, data, total, want_float=False)
    d_i = torch.zeros(total, dtype=torch.int16, device=gpu)
    afgpu.qoa_transform(len(frames), torch.from_numpy(frames.view(np.uint8).copy()).to(gpu),
                        torch.from_numpy(data).to(gpu), d_i, None)
    torch.cuda.synchronize()
    assert (d_i.cpu().numpy() == want_i).all()


def crafted(gpu, samples_per_frame, slice_words, n_frames=3, lms=None):
    """stereo frames built by hand: `slice_words(rng, n)` gives the 64-bit slices, the LMS state is random (or `lms`)"""
    import torch
    rng = np.random.default_rng(samples_per_frame)
    nsl = -(-samples_per_frame // 20)
    frames = np.zeros(n_frames, afgpu.QOA_FRAME_DTYPE)
    planes, at, out = [], 0, 0
    for f in range(n_frames):
        hdr = np.array([(2 << 56) | (44100 << 32) | (samples_per_frame << 16) | (8 + 32 + 16 * nsl)], ">u8").view(np.uint8)
        state = rng.integers(0, 256, 32, dtype=np.uint8) if lms is None else np.asarray(lms, ">i2").view(np.uint8)
        body = slice_words(rng, nsl * 2).astype(">u8").view(np.uint8)
        planes += [hdr, state, body]
        frames[f] = (at, out, samples_per_frame, 2, 0)
        at += 8 + 32 + body.size
        out += samples_per_frame * 2
    data = np.concatenate(planes)
    want_i, want_f = oraclelib.qoa_transform(frames, data, out)
    d_i = torch.full((out,), -7, dtype=torch.int16, device=gpu)
    d_f = torch.full((out,), float("nan"), dtype=torch.float32, device=gpu)
    afgpu.qoa_transform(n_frames, torch.from_numpy(frames.view(np.uint8).copy()).to(gpu), torch.from_numpy(data).to(gpu), d_i, d_f)
    torch.cuda.synchronize()
    assert (d_i.cpu().numpy() == want_i).all()
    assert (d_f.cpu().numpy().view(np.uint32) == want_f.view(np.uint32)).all()
    return want_i


@pytest.mark.parametrize("word", [0xFFFFFFFFFFFFFFFF, 0xFDB6DB6DB6DB6DB6, 0xF000000000000000, 0xFB6DB6DB6DB6DB6D])
def test_qoa_weights_at_their_largest(gpu, word):
    """The all-stereo kernel multiplies with 24-bit mads (csrc/qoa_lms.hip decode_slice): the largest scalefactor with one
    residual repeated for a whole frame drives the weights as far as a 5120-sample frame can (32767 + 5120 * 896) while the
    prediction wraps -- the bits must still be the reference's."""
    for lms in ([32767] * 4 + [32767] * 4, [-32768] * 4 + [-32768] * 4, [-32768, 32767, -32768, 32767] + [32767, -32768, 32767, -32768]):
        crafted(gpu, 5120, lambda rng, n: np.full(n, word, np.uint64), lms=lms * 2)


def test_qoa_frames_longer_than_the_format_allows(gpu):
    """a frame header may claim up to 65535 samples per channel (the reference would overrun its 5120-sample buffer; the
    oracle's holds 20480 stereo samples): past 9000 the weights can leave 24 bits and the kernel must take its 32-bit
    multiplies -- random and extreme slices over 12000- and 20000-sample stereo frames"""
    crafted(gpu, 12000, lambda rng, n: rng.integers(0, 1 << 64, n, dtype=np.uint64))
    crafted(gpu, 12000, lambda rng, n: np.full(n, 0xFFFFFFFFFFFFFFFF, np.uint64), lms=([-32768] * 4 + [32767] * 4) * 2)
    crafted(gpu, 20000, lambda rng, n: np.where(rng.random(n) < 0.5, np.uint64(0xFFFFFFFFFFFFFFFF), rng.integers(0, 1 << 64, n, dtype=np.uint64)), n_frames=2)
    want = crafted(gpu, 20000, lambda rng, n: np.full(n, 0xFFFFFFFFFFFFFFFF, np.uint64), n_frames=1, lms=([-32768] * 4 + [32767] * 4) * 2)
    assert len(np.unique(want[2 * 9400:])) > 2, "the case does not reach the weights' 24-bit limit"
