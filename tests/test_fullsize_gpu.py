"""The headline configurations at BASELINE.json's full sizes (C2: 1024 x MP3 60 s, C3: 1024 x Ogg Vorbis 2584 packets, C4:
4096 x FLAC 323 frames), one codec resident at a time: what only shows at that size -- plane offsets beyond 2^32 bytes,
every wavefront slot of the device taken, the tail of the batch -- checked through what does not need an oracle pass over
10^10 samples: the first, a middle and the LAST file against the oracle, every output written and finite, a second launch
bit-identical to the first."""
import numpy as np
import pytest

import oraclelib
from afgpu import corpus

pytestmark = pytest.mark.gpu


def free_bytes():
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()                               # what earlier tests left in torch's caching allocator
    return torch.cuda.mem_get_info()[0]


@pytest.mark.parametrize("codec,need_gb", [("mp3", 100), ("vorbis", 95), ("flac", 125)])
def test_full_size_batch(gpu, codec, need_gb):
    import torch
    if free_bytes() < need_gb * 1e9:
        pytest.skip("not enough free device memory for the full-size batch")
    wl = corpus.build_c234(gpu, 0, (codec,), corpus.C2_FILES)
    part = wl.parts[0]
    out = part.out_plane()
    out.fill_(float("nan") if out.dtype == torch.float32 else -2 ** 31)
    stream = torch.cuda.current_stream()
    part.launch(stream.cuda_stream)
    torch.cuda.synchronize()
    assert out.numel() * out.element_size() > 2 ** 32                            # the plane really is beyond 32-bit offsets
    n_files = len(part.file_bounds()) - 1
    for f in (0, n_files // 2 + 1, n_files - 1):
        r = part.check_file(oraclelib, f)
        assert r["samples"] > 0 and r["mismatches"] == 0, (codec, f, r)
    if out.dtype == torch.float32:
        assert bool(torch.isfinite(out).all())                                   # every sample written (the fill was NaN)
    first = out.clone() if free_bytes() > out.numel() * out.element_size() + (8 << 30) else None
    if first is not None:
        part.launch(stream.cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(first, out)
    del first, out, part, wl
    torch.cuda.empty_cache()


@pytest.mark.numeric_tolerance
def test_full_size_vorbis_batch_default_mode(gpu):
    """C3 in the default numeric mode (csrc/vorbis_walk.hip): first, middle and last file within 1e-5 RMS of the oracle,
    every frame written, a second launch bit-identical (work is drawn from an atomic counter: the bits must not depend on
    which wavefront took which segment)."""
    import torch
    if free_bytes() < 95e9:
        pytest.skip("not enough free device memory for the full-size batch")
    wl = corpus.build_c234(gpu, 0, ("vorbis",), corpus.C2_FILES)
    part = wl.parts[0]
    out = part.out_plane()
    out.fill_(float("nan"))
    stream = torch.cuda.current_stream()
    part.launch(stream.cuda_stream)
    torch.cuda.synchronize()
    n_files = len(part.file_bounds()) - 1
    for f in (0, n_files // 2 + 1, n_files - 1):
        r = part.check_file(oraclelib, f)
        assert r["samples"] > 0 and r["mode"] == "tolerance" and r["mismatches"] == 0 and r["rms_error"] <= 1e-5, (f, r)
        assert r["bitwise_mismatches"] > 0, "the exact kernel ran: this test would not be testing the walk"
    assert bool(torch.isfinite(out).all())
    first = out.clone()
    part.launch(stream.cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(first, out)
    del first, out, part, wl
    torch.cuda.empty_cache()


@pytest.mark.numeric_tolerance
def test_full_size_mp3_batch_default_mode(gpu):
    """C2 in the default numeric mode (mp3_tolerance_kernel: csrc/mp3_kernel.h with fused multiply-adds, the kernel the
    headline is measured on): first, middle and last file within 1e-5 RMS of the oracle, every sample written, a second
    launch bit-identical."""
    import torch
    if free_bytes() < 100e9:
        pytest.skip("not enough free device memory for the full-size batch")
    wl = corpus.build_c234(gpu, 0, ("mp3",), corpus.C2_FILES)
    part = wl.parts[0]
    out = part.out_plane()
    out.fill_(float("nan"))
    stream = torch.cuda.current_stream()
    part.launch(stream.cuda_stream)
    torch.cuda.synchronize()
    assert out.numel() * out.element_size() > 2 ** 32
    n_files = len(part.file_bounds()) - 1
    for f in (0, n_files // 2 + 1, n_files - 1):
        r = part.check_file(oraclelib, f)
        assert r["samples"] > 0 and r["mode"] == "tolerance" and r["mismatches"] == 0 and r["rms_error"] <= 1e-5, (f, r)
        assert r["bitwise_mismatches"] > 0, "the exact kernel ran: this test would not be testing the tolerance build"
    assert bool(torch.isfinite(out).all())
    first = out.clone()
    part.launch(stream.cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(first, out)
    del first, out, part, wl
    torch.cuda.empty_cache()


def test_full_size_celt_batch(gpu):
    """8192 stereo CELT streams of 200 frames (the stream-walk path, every wavefront slot taken four times over): first,
    middle and last stream against the oracle, everything written, deterministic."""
    import torch
    if free_bytes() < 60e9:
        pytest.skip("not enough free device memory")
    part = corpus.CeltPart(0xCE17, np.full(8192, 200), gpu)
    out = part.out_plane()
    out.fill_(float("nan"))
    stream = torch.cuda.current_stream()
    part.launch(stream.cuda_stream)
    torch.cuda.synchronize()
    for f in (0, 4097, 8191):
        r = part.check_file(oraclelib, f)
        assert r["samples"] == 200 * 960 * 2 and r["mismatches"] == 0, (f, r)
    assert bool(torch.isfinite(out).all())
    first = out.clone()
    part.launch(stream.cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(first, out)
    del first, out, part
    torch.cuda.empty_cache()


@pytest.mark.numeric_tolerance
def test_full_size_celt_batch_default_numeric_mode(gpu):
    """The same batch through the product's default path (csrc/celt_walk.hip: persistent segment walk, de-emphasis as a
    prefix sum): 1e-5 RMS against the oracle on the first, a middle and the last stream, < 1 % of the samples on a
    neighbouring int16, everything written, and -- items are drawn from an atomic counter -- still deterministic."""
    import torch
    if free_bytes() < 60e9:
        pytest.skip("not enough free device memory")
    part = corpus.CeltPart(0xCE17, np.full(8192, 200), gpu)
    out = part.out_plane()
    out.fill_(float("nan"))
    stream = torch.cuda.current_stream()
    part.launch(stream.cuda_stream)
    torch.cuda.synchronize()
    for f in (0, 4097, 8191):
        r = part.check_file(oraclelib, f)
        assert r["mode"] == "tolerance" and r["samples"] == 200 * 960 * 2 and r["mismatches"] == 0, (f, r)
        assert r["rms_error"] <= 1e-5 and r["int16_flip_rate"] < 0.01 and r["int16_max_step"] <= 1, (f, r)
    assert bool(torch.isfinite(out).all())
    first = out.clone()
    part.launch(stream.cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(first, out)
    del first, out, part
    torch.cuda.empty_cache()


@pytest.mark.numeric_tolerance
def test_fullsize_c5_wave(gpu):
    """BASELINE configs[4] at its real size: the first of the three resident waves of the 65 536-file mixed corpus on one GPU
    (21 846 files: MP3 + Ogg Vorbis + FLAC + Opus-CELT planes resident together, the CELT walk on a second stream beside the
    other codecs' kernels, long streams cut into segments) -- the first, a middle and the last file of every codec against
    the oracle, every output written (the planes are pre-filled with NaN / INT_MIN), a second step bit-identical."""
    import torch
    if free_bytes() < 230e9:
        pytest.skip("not enough free device memory for a full C5 wave")
    man = corpus.c5_manifest()
    waves = corpus.c5_shard_waves(man, 0, 1)
    assert len(waves) == 3 and 20000 < len(waves[0]) <= corpus.C5_WAVE_FILES
    wl = corpus.build_c5_wave(man, waves[0], gpu)
    assert [p.name for p in wl.parts] == ["mp3", "vorbis", "flac", "celt"]
    for p in wl.parts:
        o = p.out_plane()
        o.fill_(float("nan") if o.dtype == torch.float32 else -2 ** 31)
    stream = torch.cuda.current_stream()
    side = torch.cuda.Stream(device=gpu)
    wl.step(stream, None, side)
    torch.cuda.synchronize()
    for p in wl.parts:
        n_files = len(p.file_bounds()) - 1
        assert n_files == len(p.file_ids) > 1000
        for f in (0, n_files // 2 + 1, n_files - 1):
            r = p.check_file(oraclelib, f)
            assert r["samples"] > 0 and r["mismatches"] == 0, (p.name, f, r)
            if p.name == "celt":
                assert r["rms_error"] <= 1e-5 and r["int16_flip_rate"] < 0.01, (f, r)
        o = p.out_plane()
        if o.dtype == torch.float32:
            assert bool(torch.isfinite(o).all()), p.name
        else:
            assert int((o == -2 ** 31).sum()) < o.numel() // 1000, p.name           # (INT_MIN is a legal FLAC sample, not a common one)
    # the long-chain regime decides C5's time: the wave's longest Opus file (30 s = 1500 frames) is in this check
    celt = wl.parts[3]
    longest = int(np.argmax(celt.frames_per_file))
    assert celt.frames_per_file[longest] >= 1400
    r = celt.check_file(oraclelib, longest)
    assert r["mismatches"] == 0 and r["rms_error"] <= 1e-5, r
    for p in wl.parts:                                                               # determinism, one plane at a time
        o = p.out_plane()
        if free_bytes() < o.numel() * o.element_size() + (4 << 30):
            continue
        first = o.clone()
        wl.step(stream, None, side)
        torch.cuda.synchronize()
        assert torch.equal(first, o), p.name
        del first
    del wl
    torch.cuda.empty_cache()


@pytest.mark.numeric_tolerance
@pytest.mark.parametrize("ch,bs0,bs1", [(1, 256, 2048), (2, 256, 1024), (1, 256, 1024), (2, 512, 4096), (1, 512, 4096), (6, 256, 2048)])
def test_full_size_vorbis_other_shapes_default_mode(gpu, ch, bs0, bs1):
    """C3-sized batches (1024 files, the samples of 2584 stereo 2048-sample packets each: planes beyond 32-bit offsets) of
    the other stream shapes the tolerance-mode walk has kernels for: first, middle and last file within 1e-5 RMS of the
    oracle, every frame written, a second launch bit-identical."""
    import torch
    from afgpu import synthetic
    if free_bytes() < 95e9:
        pytest.skip("not enough free device memory for the full-size batch")
    files = 1024
    packets = 2584 * 2048 * 2 // (bs1 * ch)
    plan, spec = synthetic.vorbis_batch_device(0x0662 + ch + bs1, files, packets, gpu, bs0=bs0, bs1=bs1, channels=ch)
    out = torch.full((plan.out_floats,), float("nan"), dtype=torch.float32, device=gpu)
    plan.transform(spec, out)
    torch.cuda.synchronize()
    assert out.numel() * 4 > 2 ** 32
    so, oo = plan.offsets()
    for f in (0, files // 2 + 1, files - 1):
        p0, p1 = f * packets, (f + 1) * packets
        s0, o0 = int(so[p0]), int(oo[p0])
        s1 = int(so[p1]) if p1 < plan.total_packets else plan.spec_floats
        o1 = int(oo[p1]) if p1 < plan.total_packets else plan.out_floats
        want = oraclelib.vorbis_transform(plan.packets[f:f + 1], plan.channels[f:f + 1], plan.bs0[f:f + 1], plan.bs1[f:f + 1],
                                          plan.pflags[p0:p1], so[p0:p1] - so[p0], oo[p0:p1] - oo[p0], spec[s0:s1].cpu().numpy(), o1 - o0)
        got = out[o0:o1].cpu().numpy()
        rms = float(np.sqrt(np.mean((got.astype(np.float64) - want) ** 2)))
        assert rms <= 1e-5, (f, rms)
        assert (got.view(np.uint32) != want.view(np.uint32)).sum() > got.size // 10, "the exact kernel ran: this test would not be testing the walk"
    assert bool(torch.isfinite(out).all())
    first = out.clone()
    plan.transform(spec, out)
    torch.cuda.synchronize()
    assert torch.equal(first, out)
    del first, out, spec, plan
    torch.cuda.empty_cache()
