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
