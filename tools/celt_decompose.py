#!/usr/bin/env python3
"""Where the dense CELT batch's time goes: the same 8192 x 200-frame batch with and without transient frames (8 short blocks: the
general path of the walk) and live post-filters.   python tools/celt_decompose.py [--steps 5]"""
import argparse, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-formats_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=5); ap.add_argument("--streams", type=int, default=8192)
    args = ap.parse_args()
    import afgpu
    from afgpu import synthetic
    from bench_codecs import time_launches, HBM_PEAK_GBS
    dev = torch.device("cuda:0")
    for pt, pp in ((0.15, 0.3), (0.0, 0.3), (0.15, 0.0), (0.0, 0.0), (0.05, 0.6), (1.0, 0.0)):
        rb1, recs1, coef1, tot1 = synthetic.celt_batch(0x0905, [200], [2], p_transient=pt, p_postfilter=pp)
        nrec, streams = len(recs1), args.streams
        recs = np.tile(recs1, streams)
        k = np.repeat(np.arange(streams, dtype=np.uint64), nrec)
        recs["coef_off"] += k * np.uint64(coef1.size); recs["out_off"] += k * np.uint64(tot1)
        rec_base = np.concatenate([(rb1[:-1] + np.uint64(s * nrec)) for s in range(streams)] + [np.array([streams * nrec], np.uint64)])
        d_coef = torch.from_numpy(np.tile(coef1, streams)).to(dev)
        d_recs = torch.from_numpy(recs.view(np.uint8).copy()).to(dev)
        d_rb = torch.from_numpy(rec_base.view(np.int64)).to(dev)
        d_out = torch.empty(tot1 * streams, dtype=torch.float32, device=dev)
        ms = time_launches(lambda: afgpu.celt_transform(len(rec_base) - 1, d_rb, d_recs, d_coef, d_out), args.steps, 2)
        avg = sum(ms) / len(ms)
        print(json.dumps({"p_transient": pt, "p_postfilter": pp, "avg_kernel_ms": round(avg, 3), "frac": round(8 * tot1 * streams / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 3)}), flush=True)
        del d_coef, d_recs, d_rb, d_out; torch.cuda.empty_cache()

if __name__ == "__main__":
    main()
