"""Where do the FLAC kernel's samples differ from the oracle's?  (development aid)  usage: flac_mismatch_positions.py [files] [res16: 0|1]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-formats_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from afgpu import corpus
import oraclelib
files = int(sys.argv[1]) if len(sys.argv) > 1 else 4
res16 = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
host = files <= 64
p = corpus.FlacPart(1234, np.full(files, 323), torch.device("cuda:0"), host=host, res16=res16)
chk = min(files, 16)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    p.launch(s.cuda_stream)
s.synchronize()
nchk = chk * 323
cnt = nchk * 2 * p.block_size
words = cnt // 2 if res16 else cnt
want = oraclelib.flac_transform(p.frames_np[:nchk], p.sub_np[:2 * nchk], p.res[:words].cpu().numpy(), cnt)
got = p.out[:cnt].cpu().numpy()
bad = np.nonzero(got != want)[0]
print("mismatches", len(bad), "of", cnt)
if len(bad):
    fr = bad // (2 * p.block_size); t = (bad % (2 * p.block_size)) // 2; ch = bad % 2
    print("frames", np.unique(fr)[:40], "n", len(np.unique(fr)))
    print("t min/max", t.min(), t.max(), "tile steps", np.unique(t // 32)[:40])
    print("t%32", np.unique(t % 32))
    print("ch", np.unique(ch, return_counts=True))
    for b in bad[:12]:
        print(int(b), "frame", int(b // 8192), "t", int((b % 8192) // 2), "ch", int(b % 2), "got", int(got[b]), "want", int(want[b]))
