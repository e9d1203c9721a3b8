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


def test_gross_overload_follows_the_wrapping_int_arithmetic():
    """Outside the trick's working range the reference's ints wrap (D semantics): x + 384 in (-384, 0), i.e. x in
    (-768, -384), comes out as +32767, everything else beyond +-1 saturates with its sign.  Stated here with numpy's
    modular uint32 arithmetic, independently of the C restatement."""
    x = np.concatenate([np.linspace(-3000, 3000, 48001), [-768.0, -767.99994, -384.00003, -384.0, 1e30, -1e30, np.inf, -np.inf]]).astype(np.float32)
    got_i, _ = oraclelib.opus_output(x)
    t = (x + np.float32(384.0)).astype(np.float32)
    d = (t.view(np.uint32) - np.uint32(((150 - 15) << 23) + (1 << 22))).view(np.int32).astype(np.int64)
    want = np.where((d + 32768 < 0) | (d + 32768 > 65535), np.where(d < 0, -32768, 32767), d).astype(np.int16)
    assert np.array_equal(got_i, want)
    inside = (x > -768) & (x < -384.0001)
    assert (got_i[inside] == 32767).all() and inside.sum() > 1000
    assert (got_i[x <= -768] == -32768).all() and (got_i[(x > -383.9) & (x < -1.1)] == -32768).all() and (got_i[x > 1.1] == 32767).all()
