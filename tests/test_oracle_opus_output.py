"""The oracle's restatement of OpusFile.readFrame's float -> int16 conversion (dopus.d:7923-7926, :8098-8105) against
its definition: x * 32768 rounded to nearest, ties to even, saturated to int16; then / 32767.0f (stream.d:480)."""
import numpy as np

import oraclelib


def test_magic_number_rounding_is_round_half_even_of_x_times_32768():
    rng = np.random.default_rng(3)
    k = np.arange(-40000, 40000, dtype=np.float64)
    x = np.concatenate([((k + 0.5) / 32768.0), (k / 32768.0), rng.standard_normal(100000) * 0.5, [1e-9, -1e-9, 5.0, -5.0]]).astype(np.float32)
    got_i, got_f = oraclelib.opus_output(x)
    want = np.clip(np.rint(x.astype(np.float64) * 32768.0), -32768, 32767).astype(np.int16)      # rint: ties to even
    assert np.array_equal(got_i, want)
    assert np.array_equal(got_f, want.astype(np.float32) / np.float32(32767.0))
