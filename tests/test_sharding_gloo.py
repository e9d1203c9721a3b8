"""N > 1 path on CPU: two gloo ranks shard the mixed corpus (BASELINE configs[4] in miniature) by file with the
product's partition and wave logic, gather **per-file samples** on rank 0, and agree file by file with a single
process.  The product has no CPU path, so on this box the oracle stands in for the kernels; the same
test with the library doing the decoding runs on the GPU box (tests/test_multidevice_gpu.py)."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

import oraclelib
from afgpu import corpus, sharding

N_FILES = 24


def manifest():
    m = corpus.c5_manifest(N_FILES, 0x5C5)
    m["units"] = np.maximum(2, m["units"] // 64)
    per_unit = np.array([576, 1024, 4096, 960], np.int64)[m["kind"]]
    m["work"] = m["units"] * per_unit * 2
    return corpus.with_cost(m)


def decode_file(m, fid, seed=corpus.C5_SEED):
    """One corpus file from its per-file generators through the oracle (bit pattern of the output)."""
    kind, n = int(m["kind"][fid]), int(m["units"][fid])
    ids = [fid]
    if kind == corpus.KIND_MP3:
        out = oraclelib.mp3_transform([n], [2], corpus.mp3_coefs_numpy(seed, [n], ids), corpus.mp3_flag_plane(seed, [n], ids))
    elif kind == corpus.KIND_VORBIS:
        pf = corpus.vorbis_flag_plane(seed, [n], ids)
        so, oo, _, total = oraclelib.vorbis_layout(np.array([n], np.uint32), [2], [256], [2048], pf)
        out = oraclelib.vorbis_transform(np.array([n], np.uint32), np.array([2], np.uint8), np.array([256], np.uint16),
                                         np.array([2048], np.uint16), pf, so, oo, corpus.vorbis_spec_numpy(seed, pf, [n], ids), total)
    elif kind == corpus.KIND_FLAC:
        fr, sf = corpus.flac_records(seed, [n], ids)
        out = oraclelib.flac_transform(fr, sf, corpus.flac_residuals_numpy(seed, [n], ids), n * 2 * 4096)
    else:
        rb, recs, total, _ = corpus.celt_records(seed, [n], ids)
        out = oraclelib.celt_transform(rb, recs, corpus.celt_coefs_numpy(seed, [n], ids), total)
    return out.view(np.uint32).copy()


def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = manifest()
    mine = {}
    waves = corpus.c5_shard_waves(m, rank, world, wave_files=5)
    for ids in waves:
        for fid in ids:
            mine[int(fid)] = decode_file(m, int(fid))
        dist.barrier()                                # every rank has the same number of waves
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if rank == 0:
        q.put((gathered, len(waves)))
    dist.barrier()
    dist.destroy_process_group()


def test_partition_is_deterministic_and_balanced():
    m = corpus.c5_manifest()
    assert len(m["kind"]) == 65536
    frac = np.bincount(m["kind"], minlength=4) / 65536
    assert np.allclose(frac, [0.40, 0.25, 0.25, 0.10], atol=0.01)                      # SURVEY 8d mix
    assert 4.0 <= m["seconds"].min() and m["seconds"].max() <= 30.0
    assert abs(m["seconds"].mean() - 12.9) < 0.3                                       # log-uniform 4..30 s
    assert 7.0e10 < m["work"].sum() < 8.0e10                                           # ~7.5e10 samples
    assert (corpus.c5_manifest()["units"] == m["units"]).all()                         # seeds fixed
    for world in (1, 2, 4, 8):
        r = corpus.c5_partition(m, world)
        assert (r == corpus.c5_partition(m, world)).all() and r.max() < world
        waves = [corpus.c5_shard_waves(m, k, world) for k in range(world)]
        assert len({len(w) for w in waves}) == 1                                       # same wave count on every rank
        assert max(len(x) for w in waves for x in w) <= corpus.C5_WAVE_FILES
        assert sorted(np.concatenate([np.concatenate(w) for w in waves])) == list(range(65536))
        # balanced on predicted device time (samples x the codec's measured time per sample), not on samples
        assert corpus.c5_imbalance(m, world) < 1.02
        time_of = np.bincount(r, weights=m["cost"], minlength=world)
        assert time_of.max() / time_of.mean() < 1.02
        # ... and no rank collects the long Opus files: a stream whose post-filter never idles is one serial walk
        longest = m["spread"]
        assert len(longest) == 64 and (m["kind"][longest] == corpus.KIND_CELT).all()
        opus = np.flatnonzero(m["kind"] == corpus.KIND_CELT)
        assert m["units"][longest].min() >= np.sort(m["units"][opus])[-64]
        assert np.bincount(r[longest], minlength=world).max() <= -(-64 // world)          # the bound lpt_partition documents


def test_file_inputs_do_not_depend_on_their_neighbours():
    """A file's records and inputs are keyed by its id: the same whether generated alone or inside a plane."""
    seed = corpus.C5_SEED
    g = corpus.mp3_coefs_numpy(seed, [3, 5, 4], [7, 9, 11])
    assert np.array_equal(g[3 * 1152:8 * 1152], corpus.mp3_coefs_numpy(seed, [5], [9]))
    f = corpus.mp3_flag_plane(seed, [3, 5, 4], [7, 9, 11])
    assert np.array_equal(f[6:16], corpus.mp3_flag_plane(seed, [5], [9]))
    fr, sf = corpus.flac_records(seed, [2, 3], [4, 5])
    fr1, sf1 = corpus.flac_records(seed, [3], [5])
    assert np.array_equal(sf[4:], sf1) and np.array_equal(fr["assignment"][2:], fr1["assignment"])
    rb, recs, _, _ = corpus.celt_records(seed, [4, 6], [1, 2])
    rb1, recs1, _, _ = corpus.celt_records(seed, [6], [2])
    for k in ("blocks", "pf_period_new", "pf_gains_new"):
        assert np.array_equal(recs[k][8:], recs1[k])


def test_two_ranks_equal_one_process_file_by_file():
    m = manifest()
    single = {f: decode_file(m, f) for f in range(N_FILES)}
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered, n_waves = q.get(timeout=300)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert n_waves >= 2 and len(gathered) == 2
    assert sorted(list(gathered[0]) + list(gathered[1])) == list(range(N_FILES))       # a partition of the corpus
    rank_of = corpus.c5_partition(m, 2)
    for r, g in enumerate(gathered):
        assert 0 < len(g) < N_FILES
        for fid, arr in g.items():
            assert rank_of[fid] == r
            assert np.array_equal(arr, single[fid]), fid                               # per-file samples, not a checksum
