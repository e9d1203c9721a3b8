"""ns per decoded sample of the MP3 and Vorbis walks against file length and against files per launch (VERDICT r04
item 2: the C5 members run 10-13 % slower than C2 / C3).  Same total work per row where memory allows.
usage: python tools/c5_lengths.py [out.json]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-formats_amd"))
import torch
from afgpu import corpus

dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev)


def time_part(p, reps=6):
    with torch.cuda.stream(stream):
        p.launch(stream.cuda_stream)
        stream.synchronize()
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream); p.launch(stream.cuda_stream); b.record(stream); b.synchronize()
            ts.append(a.elapsed_time(b))
    return float(np.median(ts))


rows = []
GR_PER_S, PK_PER_S = 2 * 44100 / 1152.0, 44100 / 1024.0
total_gr = 1024 * corpus.C2_GRANULES // 2          # half of C2 per row: room for every length
total_pk = 1024 * corpus.C3_PACKETS // 2
for secs in (2, 4, 8, 15, 30, 60):
    g = int(round(secs * GR_PER_S)) & ~1
    n = total_gr // g
    p = corpus.Mp3Part(11, np.full(n, g), dev)
    ms = time_part(p)
    rows.append({"codec": "mp3", "seconds": secs, "files": n, "units_per_file": g, "samples": p.samples, "ms": ms, "ns_per_sample": ms * 1e6 / p.samples})
    print(rows[-1], flush=True)
    del p; torch.cuda.empty_cache()
    k = int(round(secs * PK_PER_S))
    n = total_pk // k
    p = corpus.VorbisPart(12, np.full(n, k), dev)
    ms = time_part(p)
    rows.append({"codec": "vorbis", "seconds": secs, "files": n, "units_per_file": k, "samples": p.samples, "ms": ms, "ns_per_sample": ms * 1e6 / p.samples})
    print(rows[-1], flush=True)
    del p; torch.cuda.empty_cache()
# mixed lengths as C5 draws them (4 .. 30 s uniform) and files per launch at a fixed length
rng = np.random.default_rng(5)
for frac in (1.0, 0.5, 0.25, 0.125):
    secs = rng.uniform(4, 30, int(3600 * frac))
    p = corpus.Mp3Part(13, (np.round(secs * GR_PER_S).astype(np.int64) & ~1), dev)
    ms = time_part(p)
    rows.append({"codec": "mp3", "seconds": "4-30 uniform", "files": len(secs), "samples": p.samples, "ms": ms, "ns_per_sample": ms * 1e6 / p.samples})
    print(rows[-1], flush=True)
    del p; torch.cuda.empty_cache()
    p = corpus.VorbisPart(14, np.round(secs * PK_PER_S).astype(np.int64), dev)
    ms = time_part(p)
    rows.append({"codec": "vorbis", "seconds": "4-30 uniform", "files": len(secs), "samples": p.samples, "ms": ms, "ns_per_sample": ms * 1e6 / p.samples})
    print(rows[-1], flush=True)
    del p; torch.cuda.empty_cache()
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/c5_lengths.json"
json.dump({"rows": rows}, open(out, "w"), indent=1)
