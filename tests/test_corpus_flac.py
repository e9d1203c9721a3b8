"""The C4 / C5 FLAC generator (afgpu/corpus.py) as SURVEY 8d specifies it: quantised LPC of a STABLE AR process, Laplacian
residuals, so that every decoded sample stays inside 17 bits like a real 16-bit file's -- checked on the oracle (CPU)."""
import numpy as np

import oraclelib
from afgpu import corpus, synthetic


def test_pool_filters_are_stable_with_bounded_gain():
    pool = corpus._lpc_pool(0xF1AC)
    for order in (8, 12):
        for coef, shift in pool[order]:
            assert len(coef) == order and 0 <= shift <= 15
            a = coef.astype(np.float64) / (1 << shift)
            roots = np.roots(np.concatenate([[1.0], -a]))
            assert np.abs(roots).max() < 0.97, "a pole of the quantised synthesis filter at or outside the unit circle"


def test_generated_files_decode_inside_17_bits():
    fpf = np.array([3, 2, 4, 1])
    frames, subs = corpus.flac_records(0xF1AC, fpf, block_size=4096)
    res = corpus.flac_residuals_numpy(0xF1AC, fpf, block_size=4096)
    assert np.abs(res).max() <= synthetic.FLAC_C4_RESIDUAL_CLAMP
    out = oraclelib.flac_transform(frames, subs, res, res.size)
    pcm = out.astype(np.int64) >> 16                                   # 16-bit samples, left-justified by drflac_read_s32
    assert np.abs(pcm).max() < (1 << 16)
    assert (out & 0xffff == 0).all()
    # worst case of the bound: every residual at the clamp with the sign that excites the filter
    worst = np.where(res >= 0, synthetic.FLAC_C4_RESIDUAL_CLAMP, -synthetic.FLAC_C4_RESIDUAL_CLAMP).astype(np.int32)
    out = oraclelib.flac_transform(frames, subs, worst, worst.size)
    assert np.abs(out.astype(np.int64) >> 16).max() < (1 << 16)
