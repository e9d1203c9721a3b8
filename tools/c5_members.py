"""The C5 wave's MP3 / Vorbis members: timed inside the resident wave (all four parts allocated), then each alone on an
otherwise empty device with the same files.  usage: python tools/c5_members.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-formats_amd"))
import torch
from afgpu import corpus
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev)

def time_part(p, reps=5):
    with torch.cuda.stream(stream):
        p.launch(stream.cuda_stream); stream.synchronize()
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream); p.launch(stream.cuda_stream); b.record(stream); b.synchronize()
            ts.append(a.elapsed_time(b))
    return float(np.median(ts))

man = corpus.c5_manifest(65536)
waves = corpus.c5_shard_waves(man, 0, 1, corpus.C5_WAVE_FILES)
ids = waves[0]
wl = corpus.build_c5_wave(man, ids, dev)
print("free GB with the wave resident", torch.cuda.mem_get_info()[0] / 1e9)
res = {}
for p in wl.parts:
    if p.name in ("mp3", "vorbis", "flac"):
        ms = time_part(p)
        res[p.name] = {"in_wave_ms": ms, "samples": p.samples, "in_wave_ps": ms * 1e9 / p.samples, "files": len(p.file_ids)}
        print(p.name, res[p.name], flush=True)
units = {p.name: (np.array(p.granules) if p.name == "mp3" else None) for p in wl.parts}
mp3_g = [np.array(p.granules) for p in wl.parts if p.name == "mp3"][0]
vb_n = [np.array(p.plan.packets) for p in wl.parts if p.name == "vorbis"][0]
del wl, p
torch.cuda.empty_cache()
print("free GB empty", torch.cuda.mem_get_info()[0] / 1e9)
p = corpus.Mp3Part(21, mp3_g, dev); ms = time_part(p); print("mp3 alone", ms, ms * 1e9 / p.samples); res["mp3"]["alone_ps"] = ms * 1e9 / p.samples
# the same files, sorted by length (longest first)
del p; torch.cuda.empty_cache()
p = corpus.Mp3Part(21, np.sort(mp3_g)[::-1].copy(), dev); ms = time_part(p); print("mp3 alone, longest first", ms, ms * 1e9 / p.samples)
del p; torch.cuda.empty_cache()
half = mp3_g[: len(mp3_g) // 2]
p = corpus.Mp3Part(21, half, dev); ms = time_part(p); print("mp3 alone, half the files", ms, ms * 1e9 / p.samples)
del p; torch.cuda.empty_cache()
p = corpus.VorbisPart(22, vb_n, dev); ms = time_part(p); print("vorbis alone", ms, ms * 1e9 / p.samples); res["vorbis"]["alone_ps"] = ms * 1e9 / p.samples
del p; torch.cuda.empty_cache()
p = corpus.VorbisPart(22, vb_n[: len(vb_n) // 2], dev); ms = time_part(p); print("vorbis alone, half the files", ms, ms * 1e9 / p.samples)
json.dump(res, open("gpurun_out/c5_members.json", "w"), indent=1)
