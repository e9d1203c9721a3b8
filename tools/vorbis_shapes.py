#!/usr/bin/env python3
"""C3-sized Vorbis batches of other stream shapes (mono, blocksize_1 1024 / 4096) through afg_vorbis_transform_hip:
kernel time, fraction of 8 TB/s on the algorithmic bytes and the error of the first file against the oracle.

    python tools/vorbis_shapes.py [--files 1024] [--steps 5] [--out profiles/r05_vorbis_shapes.json]
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

SHAPES = [(2, 256, 2048), (1, 256, 2048), (2, 256, 1024), (1, 256, 1024), (2, 512, 4096), (1, 512, 4096), (2, 256, 4096),
          (6, 256, 2048), (3, 256, 1024), (6, 512, 4096)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
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
    for ch, bs0, bs1 in [SHAPES[i] for i in only]:
        packets = 2584 * 2048 * 2 // (bs1 * ch)          # the samples of a C3 file
        plan, spec = synthetic.vorbis_batch_device(0x0662, args.files, packets, dev, bs0=bs0, bs1=bs1, channels=ch)
        out = torch.empty(plan.out_floats, dtype=torch.float32, device=dev)
        ms = time_launches(lambda: plan.transform(spec, out), args.steps, args.warmup)
        avg = sum(ms) / len(ms) * 1e-3
        alg = 4 * plan.spec_floats + plan.total_packets + 4 * plan.out_floats
        so, oo = plan.offsets()
        npk = int(plan.packets[0])
        s_end = int(so[npk]) if plan.total_packets > npk else plan.spec_floats
        o_end = int(oo[npk]) if plan.total_packets > npk else plan.out_floats
        want = oraclelib.vorbis_transform(plan.packets[:1], plan.channels[:1], plan.bs0[:1], plan.bs1[:1],
                                          plan.pflags[:npk], so[:npk], oo[:npk], spec[:s_end].cpu().numpy(), o_end)
        got = out[:o_end].cpu().numpy()
        row = {"channels": ch, "blocksize_0": bs0, "blocksize_1": bs1, "files": args.files, "packets": packets,
               "samples_per_step": plan.out_floats, "avg_kernel_ms": avg * 1e3, "samples_per_s": plan.out_floats / avg,
               "frac": alg / avg / 1e9 / HBM_PEAK_GBS, "numeric_mode": ("exact", "tolerance")[afgpu.get_numeric_mode()],
               "bitwise_mismatches": int((got.view(np.uint32) != want.view(np.uint32)).sum()),
               "rms_error": float(np.sqrt(np.mean((got.astype(np.float64) - want) ** 2))),
               "rms_signal": float(np.sqrt(np.mean(want.astype(np.float64) ** 2)))}
        print(json.dumps(row), flush=True)
        rows.append(row)
        del plan, spec, out
        torch.cuda.empty_cache()
    if args.out:
        with open(args.out, "w") as f:
            json.dump({"shapes": rows}, f, indent=1)
    bad = [f"{r['channels']}ch {r['blocksize_1']}" for r in rows
           if not r["rms_error"] <= 1e-5 or (r["numeric_mode"] == "exact" and r["bitwise_mismatches"])]
    print(json.dumps({"vorbis_shapes": {"workload": f"{args.files} files per shape, the samples of 2584 stereo 2048-sample packets each",
                                        "shapes": [{k: r[k] for k in ("channels", "blocksize_0", "blocksize_1", "samples_per_step", "avg_kernel_ms",
                                                                     "samples_per_s", "frac", "numeric_mode", "rms_error", "rms_signal")} for r in rows],
                                        "error": ("parity: " + ", ".join(bad)) if bad else None}}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
