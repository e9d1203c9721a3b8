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
