"""development: a step with its parts on one stream (as bench.py times it) against the parts side by side on several streams;
c234 at full size and the first wave of C5"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "audio-formats_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
from afgpu import corpus

dev = torch.device("cuda:0")
main = torch.cuda.Stream(device=dev)
side = torch.cuda.Stream(device=dev)
lanes = [torch.cuda.Stream(device=dev) for _ in range(4)]


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def run(tag, wl):
    print(tag, "serial", round(timed(lambda: wl.step(main, None, side)), 3))
    names = [p.name for p in wl.parts]
    import itertools
    orders = list(itertools.permutations(range(len(names)))) if len(names) <= 3 else [list(range(len(names))), list(reversed(range(len(names))))]
    for nl in (3,):
        for o in orders:
            print(tag, nl, "lanes", [names[i] for i in o], round(timed(lambda: wl.step_side_by_side(main, lanes[:nl], side, None, o)), 3))
    print(tag, "serial again", round(timed(lambda: wl.step(main, None, side)), 3))
    import oraclelib
    wl.step_side_by_side(main, lanes[:3], side)
    torch.cuda.synchronize()
    print(tag, "parity after a side-by-side step", {p.name: p.check(oraclelib).get("mismatches") for p in wl.parts})


which = sys.argv[1] if len(sys.argv) > 1 else "both"
if which in ("c234", "both"):
    wl = corpus.build_c234(dev)
    run("c234", wl)
    del wl
    torch.cuda.empty_cache()
if which in ("c5", "both"):
    man = corpus.c5_manifest(65536)
    waves = corpus.c5_shard_waves(man, 0, 1, corpus.C5_WAVE_FILES)
    wl = corpus.build_c5_wave(man, waves[0], dev)
    run("c5 wave 0", wl)
