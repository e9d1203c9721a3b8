#!/usr/bin/env python3
"""tools/ubench_pcie.py -- host<->device copy rates with page-locked memory: one copy at a time against several in flight on
separate streams (do the SDMA engines add up?), both directions alone and together.  Development aid for the batch path."""
import time
import torch

dev = torch.device("cuda:0")
N = 1 << 30                                     # bytes per direction and test


def run(label, n_streams, h2d, d2h):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    part = N // n_streams
    hs = [torch.empty(part, dtype=torch.uint8).pin_memory() for _ in range(n_streams)]
    hd = [torch.empty(part, dtype=torch.uint8).pin_memory() for _ in range(n_streams)]
    ds = [torch.empty(part, dtype=torch.uint8, device=dev) for _ in range(n_streams)]
    dd = [torch.empty(part, dtype=torch.uint8, device=dev) for _ in range(n_streams)]
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k, s in enumerate(streams):
            with torch.cuda.stream(s):
                if h2d:
                    ds[k].copy_(hs[k], non_blocking=True)
                if d2h:
                    hd[k].copy_(dd[k], non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    total = N * (int(h2d) + int(d2h))
    print(f"{label:34s} {n_streams} stream(s): {total / best / 1e9:6.1f} GB/s total")


for n in (1, 2, 4, 8):
    run("H2D", n, True, False)
for n in (1, 2, 4, 8):
    run("D2H", n, False, True)
for n in (1, 2, 4):
    run("H2D + D2H together", n, True, True)
