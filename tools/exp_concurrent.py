"""Experiment: do the three codecs of the headline workload finish sooner when their kernels are queued on three streams
at once (different units saturate: MP3 issues, FLAC / Vorbis wait for memory) than back to back on one?"""
import sys, time
sys.path.insert(0, "audio-formats_amd")
import torch
from afgpu import corpus
dev = torch.device("cuda:0")
wl = corpus.build_c234(dev)
main = torch.cuda.Stream()
side = [torch.cuda.Stream() for _ in wl.parts]
def run(mode, steps=6):
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        if mode == "serial":
            wl.step(main)
        else:
            fork = torch.cuda.Event(); fork.record(main)
            order = list(range(len(wl.parts)))
            if mode == "concurrent_flac_first":
                order = order[::-1]
            for i in order:
                side[i].wait_event(fork)
                wl.parts[i].launch(side[i])
                e = torch.cuda.Event(); e.record(side[i]); main.wait_event(e)
    torch.cuda.synchronize()
    return (time.time() - t0) / steps * 1e3
for mode in ("serial", "concurrent", "concurrent_flac_first", "serial"):
    run(mode, 2)
    print(mode, "%.2f ms per step" % run(mode))
