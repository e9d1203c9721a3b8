#!/usr/bin/env python3
"""transcode input.{mp3|flac|ogg|qoa} output.{wav|qoa} -- the reference's examples/transcode flow (main.d:12-84) over the
device library: open, report format / rate / channels / length, read 1024-frame chunks, write them out with the
library's own writers (afg_wav_encode / afg_qoa_encode_hip).  The WAV output is
32-bit float WAV (the reference example writes 24-bit PCM with TPDF dither driven by libc rand(), whose bytes are
not reproducible; BASELINE config C1 is checked on the decoded floats)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-formats_amd"))

import numpy as np  # noqa: E402


def main(argv):
    if len(argv) != 3:
        print("usage: transcode input.{mp3|flac|ogg|qoa} output.{wav|qoa}")
        return 2
    import afgpu
    data = open(argv[1], "rb").read()
    s = afgpu.AudioStream()
    s.openFromMemory(data)
    if s.isError():
        print(s.errorMessage())
        return 1
    rate, ch, length = s.getSamplerate(), s.getNumChannels(), s.getLengthInFrames()
    print(f"Opening {argv[1]}:")
    print(f"  * format     = {afgpu.FORMAT_NAMES[s.getFormat()]}")
    print(f"  * samplerate = {rate:g} Hz")
    print(f"  * channels   = {ch}")
    print("  * length     = unknown" if length == afgpu.UNKNOWN_LENGTH
          else f"  * length     = {length / rate:.3g} seconds ({length} frames)")
    buf = np.empty(1024 * ch, np.float32)
    chunks, total = [], 0
    while True:
        n = s.readSamplesFloat(buf)
        if s.isError():
            print(s.errorMessage())
            return 1
        if n <= 0:
            break
        chunks.append(buf[:n * ch].copy())
        total += n
    pcm = np.concatenate(chunks) if chunks else np.zeros(0, np.float32)
    pcm = pcm.reshape(-1, max(1, ch))
    if argv[2].lower().endswith(".qoa"):                 # the library's encoder (afg_qoa_encode_hip), float input
        import torch
        recs, _, n_out = afgpu.qoa_encode_layout([pcm.shape], int(rate))
        dev = torch.device("cuda:0")
        d_out = torch.zeros(n_out, dtype=torch.uint8, device=dev)
        afgpu.qoa_encode(1, torch.from_numpy(recs.view(np.uint8).copy()).to(dev), d_out,
                         d_pcm_f32=torch.from_numpy(np.clip(pcm, -1, 1).reshape(-1).copy()).to(dev))
        torch.cuda.synchronize()
        out = d_out.cpu().numpy()[:afgpu.qoa_encoded_size(len(pcm), ch)].tobytes()
    else:                                                # the library's WAV writer (afg_wav_encode), 32-bit float
        out = afgpu.wav_encode(pcm, int(rate), afgpu.WAV_FP32LE)
    with open(argv[2], "wb") as fh:
        fh.write(out)
    print(f"=> {total} frames decoded and written to {argv[2]}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
