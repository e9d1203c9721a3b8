"""The three throughput kernels launched side by side on three streams (corpus.Workload.step_side_by_side, what
bench.py reports as other_workloads.c234_side_by_side): the persistent MP3 / Vorbis grids and FLAC's grid share the device,
and every one of them must write the bits it writes when it has the device to itself."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def planes(wl):
    return [p.out_plane().cpu().numpy().copy() for p in wl.parts]


def run_case(gpu):
    import torch
    from afgpu import corpus
    wl = corpus.build_c234(gpu, files=24)
    main = torch.cuda.Stream(device=gpu)
    lanes = [torch.cuda.Stream(device=gpu) for _ in wl.parts]
    for p in wl.parts:
        p.out_plane().zero_()
    torch.cuda.synchronize()
    wl.step(main)
    torch.cuda.synchronize()
    want = planes(wl)
    for order in ([1, 0, 2], [2, 1, 0], [0, 2, 1]):
        for p in wl.parts:
            p.out_plane().fill_(float("nan") if p.out_plane().dtype.is_floating_point else -7)
        torch.cuda.synchronize()
        for _ in range(3):                                           # back to back: the per-launch counters rotate
            wl.step_side_by_side(main, lanes, None, None, order)
        torch.cuda.synchronize()
        for p, w, g in zip(wl.parts, want, planes(wl)):
            assert np.array_equal(g.view(np.uint32), w.view(np.uint32)), (p.name, order)
    import oraclelib
    for p in wl.parts:
        assert p.check(oraclelib, 1)["mismatches"] == 0, p.name


def test_side_by_side_exact_mode(gpu):
    run_case(gpu)


@pytest.mark.numeric_tolerance
def test_side_by_side_default_mode(gpu):
    run_case(gpu)
