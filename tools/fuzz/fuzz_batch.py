#!/usr/bin/env python3
"""Mutation fuzzing of the whole batch path on a device: damaged MP3 (Layer I-III) / Ogg Vorbis / Ogg Opus / FLAC / QOA files through
afg_batch_decode in mixed batches.  Every item must come back (status ok or an error message, never a crash), finite,
and identical to what the same bytes give when decoded on their own (the staged and the per-file paths agree).
usage: python tools/fuzz/fuzz_batch.py [batches] [seed]      (run it under `timeout`)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "audio-formats_amd"), os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402


def mutate(rng, data):
    v = bytearray(data)
    for _ in range(int(rng.integers(1, 6))):
        pos = int(rng.integers(0, len(v)))
        kind = int(rng.integers(0, 5))
        if kind == 0:
            v[pos] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            v[pos] = int(rng.integers(0, 256))
        elif kind == 2:
            del v[pos:pos + int(rng.integers(1, 300))]
        elif kind == 3:
            v[pos:pos] = bytes(rng.integers(0, 256, int(rng.integers(1, 80)), dtype=np.uint8))
        else:
            del v[pos:]                                      # truncation
        if not v:
            v = bytearray(b"\0")
    return bytes(v)


def main():
    import afgpu
    import flac_bitstream as fb
    import mp3_bitstream as mb
    import mp3_l12_bitstream as lb
    import opus_bitstream as ob
    import oraclelib
    import vorbis_bitstream as vb
    from test_flac_frontend import make_pcm
    batches = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    gold = os.path.join(ROOT, "tests", "golden")
    seeds = [open(os.path.join(gold, "mathjax_invalid_keypress.mp3"), "rb").read(),
             open(os.path.join(gold, "mathjax_invalid_keypress.ogg"), "rb").read(),
             mb.make_file(3, n_frames=12, version="mpeg1", sr=0, mode="ms", bitrate_index=9)[0],
             mb.make_file(4, n_frames=10, version="mpeg2", sr=1, mode="mono", bitrate_index=6)[0],
             vb.make_file(5, n_packets=20), vb.make_file(6, n_packets=12, channels=1),
             fb.encode_file(make_pcm(4096 + 300, 2, 16, 3), 16, 1024, orders=(8, 12, 3))[0],
             fb.encode_file(make_pcm(1152 * 2, 1, 24, 4), 24, 1152)[0],
             oraclelib.qoa_encode(make_pcm(5120 + 777, 2, 16, 5).astype(np.int16), 44100)[0].tobytes(),
             ob.random_celt_file(np.random.default_rng(7), 2, 25, preskip=0)[0],
             ob.random_celt_file(np.random.default_rng(8), 1, 18, preskip=0, gain=500)[0],
             lb.random_file(np.random.default_rng(9), 2, 16, vary_bitrate=True),
             lb.random_file(np.random.default_rng(10), 1, 30, mode="mono", vary_bitrate=True)]
    ok = bad = 0
    for b in range(batches):
        files = [mutate(rng, seeds[int(rng.integers(0, len(seeds)))]) if rng.random() < 0.8 else seeds[int(rng.integers(0, len(seeds)))]
                 for _ in range(48)]
        got = afgpu.batch_decode(files, n_threads=8)
        assert len(got) == len(files)
        pick = rng.choice(len(files), 6, replace=False)
        for i, g in enumerate(got):
            if g["status"] == 0:
                ok += 1
                assert g["frames"] >= 0 and (g["pcm"] is None or np.isfinite(g["pcm"]).all()), (b, i)
            else:
                bad += 1
                assert g["message"], (b, i)
            if i in pick:                                    # the same bytes alone: same verdict, same samples
                a = afgpu.batch_decode([files[i]], n_threads=1)[0]
                assert (a["status"], a["frames"], a["channels"]) == (g["status"], g["frames"], g["channels"]), (b, i)
                if a["pcm"] is not None:
                    assert np.array_equal(a["pcm"].view(np.uint32), g["pcm"].view(np.uint32)), (b, i)
    print(f"ok: {batches} batches, {ok} files decoded, {bad} rejected")


if __name__ == "__main__":
    main()
