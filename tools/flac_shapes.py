#!/usr/bin/env python3
"""C4-sized FLAC batches of other stream shapes through afg_flac_transform_variants_hip: kernel time, fraction of 8 TB/s on the
bytes moved, and the first frames against the oracle (bit for bit).  A prototype of a few hundred frames is generated on the
host (afgpu.synthetic.flac_batch) and repeated on the device -- the kernel's time does not depend on the values.

    python tools/flac_shapes.py [--samples 1.08e10] [--steps 5] [--out profiles/r06_flac_shapes.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-formats_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# label, channels, bits per sample, LPC orders, block size, int16 residual rows
SHAPES = [("2ch/16bit/order8+12/4096 int16 rows (C4)", 2, 16, (8, 12), 4096, True),
          ("2ch/16bit/order8+12/4096 int32 rows", 2, 16, (8, 12), 4096, False),
          ("2ch/24bit/order12/4096 (wide sums, int32 rows)", 2, 24, (12,), 4096, False),
          ("2ch/16bit/order32/4096 int16 rows", 2, 16, (32,), 4096, True),
          ("2ch/24bit/order32/4096 (wide sums, int32 rows)", 2, 24, (32,), 4096, False),
          ("2ch/16bit/order4/4096 int16 rows", 2, 16, (2, 4), 4096, True),
          ("1ch/16bit/order8+12/4096 int16 rows", 1, 16, (8, 12), 4096, True),
          ("6ch/16bit/order8+12/4096 int16 rows", 6, 16, (8, 12), 4096, True),
          ("2ch/16bit/order8+12/1152 int16 rows", 2, 16, (8, 12), 1152, True),
          ("2ch/16bit/order8+12/576 int16 rows", 2, 16, (8, 12), 576, True)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=float, default=1.0838e10, help="samples per batch (C4: 4096 files x 323 frames x 8192)")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--proto-frames", type=int, default=256)
    ap.add_argument("--out", default="")
    ap.add_argument("--only", default="", help="comma-separated indices into SHAPES")
    args = ap.parse_args()
    import afgpu
    import oraclelib
    from afgpu import synthetic
    from bench_codecs import time_launches, HBM_PEAK_GBS
    oraclelib.build()
    dev = torch.device("cuda:0")
    rows = []
    only = [int(x) for x in args.only.split(",")] if args.only else range(len(SHAPES))
    for label, ch, bps, orders, bs, rows16 in [SHAPES[i] for i in only]:
        pf = args.proto_frames
        frames, sub, res, total = synthetic.flac_batch(0xF1AC + bs + ch, pf, block_size=bs, channels=ch, bps=bps, orders=orders,
                                                       residual_scale=24.0, wasted_p=0.02)
        if rows16:
            res = np.clip(res, -30000, 30000).astype(np.int32)
            frames, res = synthetic.flac_pack16(frames, res)
            assert (frames["res16"] != 0).all()
        res = np.concatenate([res, np.zeros((-len(res)) % 4, np.int32)])      # copies start 16-byte aligned
        want = oraclelib.flac_transform(frames, sub, res, total)
        words = len(res)                                  # int32 words of one prototype's residual plane
        reps = max(1, int(round(args.samples / total)))
        # the batch: `reps` copies of the prototype, records shifted copy by copy
        fr = np.tile(frames, reps)
        k = np.repeat(np.arange(reps, dtype=np.uint64), pf)
        in_scale = 2 if rows16 else 1                     # in_off counts int16 elements of packed rows
        fr["in_off"] += k * np.uint64(words * in_scale)
        fr["out_off"] += k * np.uint64(total)
        fr["sf_index"] += (k * np.uint64(pf * ch)).astype(np.uint32)
        sb = np.tile(sub, reps)
        d_res = torch.from_numpy(res).to(dev).repeat(reps)
        d_frames = torch.from_numpy(fr.view(np.uint8).copy()).to(dev)
        d_sub = torch.from_numpy(sb.view(np.uint8).copy()).to(dev)
        d_out = torch.empty(total * reps, dtype=torch.int32, device=dev)
        variants = afgpu.flac_variants(fr, sb)
        n = len(fr)
        ms = time_launches(lambda: afgpu.flac_transform(n, d_frames, d_sub, d_res, d_out, None, None, variants=variants), args.steps, args.warmup)
        avg = sum(ms) / len(ms) * 1e-3
        got_first = d_out[:total].cpu().numpy()
        got_last = d_out[-total:].cpu().numpy()
        samples = total * reps
        moved = (2 if rows16 else 4) * samples + 4 * samples + n * 32 + len(sb) * 68
        row = {"label": label, "channels": ch, "bps": bps, "orders": list(orders), "block_size": bs, "int16_rows": rows16, "frames": n,
               "samples_per_step": samples, "avg_kernel_ms": avg * 1e3, "samples_per_s": samples / avg,
               "bytes_moved": moved, "frac": moved / avg / 1e9 / HBM_PEAK_GBS, "variants_mask": variants,
               "mismatches": int((got_first != want).sum()) + int((got_last != want).sum())}
        print(json.dumps(row), flush=True)
        rows.append(row)
        del d_res, d_frames, d_sub, d_out
        torch.cuda.empty_cache()
    rec = {"flac_shapes": {"workload": "C4-sized batches (1.08e10 samples) of other FLAC stream shapes, a 256-frame prototype repeated on the device",
                           "shapes": rows, "error": None}}
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(rec, fh, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
