"""The D source leaves the evaluation precision of its table expressions to the compiler / Phobos release
(VERDICT r1 weak #1): `cos(4*k*M_PI/n)` with a float `M_PI` (stb_vorbis2.d:652, :857-872) may run as cosf, cos or the
x87 `real` routine, with the angle rounded to float or kept in extended precision; CELT's tables use `real`
arithmetic (dopus.d:1489-1499), which is 80-bit on x86 and plain double elsewhere.  The oracle (and the product)
pick one reading each; these tests rebuild the tables under every other legal reading, run the same transform
stage, and require the PCM to stay within the north-star tolerance (1e-5 RMS on the API scale, full scale = 1.0) of
the chosen reading -- "bit-exact to our reading" then implies "within tolerance of every reading"."""
import numpy as np
import pytest

import oraclelib
from afgpu import corpus, synthetic

TOL = 1e-5


@pytest.fixture
def table_modes():
    L = oraclelib.lib()
    yield L
    L.afgo_vorbis_set_table_mode(0)
    L.afgo_celt_set_table_mode(0)


def vorbis_workload(amplitude):
    """C3 in miniature: 3 stereo streams, block sizes 2048/256 with short runs, floor-curve-shaped spectra."""
    packets, ch = [40, 28, 33], [2, 2, 1]
    bs0, bs1 = [256] * 3, [2048] * 3
    pflags, spec = synthetic.vorbis_batch(0x0662, packets, ch, bs0, bs1, p_short_run=0.1, amplitude=amplitude)
    so, oo, _, total = oraclelib.vorbis_layout(np.array(packets, np.uint32), ch, bs0, bs1, pflags)
    return lambda: oraclelib.vorbis_transform(packets, ch, bs0, bs1, pflags, so, oo, spec, total)


def test_vorbis_tables_differ_but_pcm_stays_within_tolerance(table_modes):
    L = table_modes
    run = vorbis_workload(1.0)
    L.afgo_vorbis_set_table_mode(0)
    ref_raw = run()
    # scale the input so that the chosen reading's output is loud, full-scale music (RMS 0.25, peaks near 1.0)
    amp = 0.25 / float(np.sqrt(np.mean(ref_raw.astype(np.float64) ** 2)))
    run = vorbis_workload(amp)
    ref = run().astype(np.float64)
    assert 0.2 < np.sqrt(np.mean(ref ** 2)) < 0.3 and np.abs(ref).max() < 2.0
    base_tabs = oraclelib.vorbis_tables(2048)
    worst = 0.0
    for mode, name in ((1, "cosf on the float angle"), (2, "angle and cos in x87 real"), (3, "float angle, cos in real")):
        L.afgo_vorbis_set_table_mode(mode)
        tabs = oraclelib.vorbis_tables(2048)
        got = run().astype(np.float64)
        rms = float(np.sqrt(np.mean((got - ref) ** 2)))
        worst = max(worst, rms)
        print(f"vorbis reading {mode} ({name}): tables differ in {int((tabs['A'] != base_tabs['A']).sum())} of 1024 A entries, "
              f"{int((tabs['window'] != base_tabs['window']).sum())} of 1024 window entries; PCM rms diff {rms:.3e}, max {np.abs(got - ref).max():.3e}")
        assert rms <= TOL, (name, rms)
    L.afgo_vorbis_set_table_mode(0)
    assert np.array_equal(run().astype(np.float64), ref)                 # the switch restores the chosen reading
    assert worst > 0.0                                                   # the readings are really different tables


def test_celt_tables_in_double_stay_within_tolerance(table_modes):
    L = table_modes
    fps = [30, 22, 17]
    rb, recs, total, _ = corpus.celt_records(0x0905, fps)
    coef = corpus.celt_coefs_numpy(0x0905, fps)
    L.afgo_celt_set_table_mode(0)
    raw = oraclelib.celt_transform(rb, recs, coef, total).astype(np.float64)
    amp = 0.25 / float(np.sqrt(np.mean(raw ** 2)))                      # API scale: ff_celt_decode_frame's floats, full scale 1.0
    coef = (coef * np.float32(amp)).astype(np.float32)
    ref = oraclelib.celt_transform(rb, recs, coef, total)
    L.afgo_celt_set_table_mode(1)
    got = oraclelib.celt_transform(rb, recs, coef, total)
    d = got.astype(np.float64) - ref.astype(np.float64)
    rms = float(np.sqrt(np.mean(d ** 2)))
    print(f"celt tables in double: PCM rms diff {rms:.3e}, max {np.abs(d).max():.3e}, "
          f"{float((got.view(np.uint32) != ref.view(np.uint32)).mean()):.3f} of the samples differ in the last bits")
    assert rms <= TOL
    # after OpusFile.readFrame's int16 rounding (dopus.d:8098-8105) a last-ulp difference can flip a sample by 1/32767
    qi_ref, qf_ref = oraclelib.opus_output(ref)
    qi_got, qf_got = oraclelib.opus_output(got)
    flips = float((qi_ref != qi_got).mean())
    qrms = float(np.sqrt(np.mean((qf_ref.astype(np.float64) - qf_got) ** 2)))
    print(f"  after int16 rounding: {flips:.4%} of the samples flip by one step, rms {qrms:.3e}")
    assert np.abs(qi_ref.astype(np.int32) - qi_got).max() <= 1 and flips < 0.10 and qrms <= TOL
