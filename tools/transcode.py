#!/usr/bin/env python3
"""transcode input.{mp3|flac|qoa} output.wav -- the reference's examples/transcode flow (main.d:12-84) over the device
library: open, report format / rate / channels / length, read 1024-frame chunks, write them out.  The output is
32-bit float WAV (the reference example writes 24-bit PCM with TPDF dither driven by libc rand(), whose bytes are
not reproducible; BASELINE config C1 is checked on the decoded floats)."""
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-formats_amd"))

import numpy as np  # noqa: E402


def main(argv):
    if len(argv) != 3:
        print("usage: transcode input.{mp3|flac|qoa} output.wav")
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
    with open(argv[2], "wb") as fh:                    # WAVE_FORMAT_IEEE_FLOAT
        fh.write(b"RIFF" + struct.pack("<I", 4 + 26 + 12 + 8 + pcm.nbytes) + b"WAVE")
        fh.write(b"fmt " + struct.pack("<IHHIIHHH", 18, 3, ch, int(rate), int(rate) * ch * 4, ch * 4, 32, 0))
        fh.write(b"fact" + struct.pack("<II", 4, total))
        fh.write(b"data" + struct.pack("<I", pcm.nbytes))
        fh.write(pcm.astype("<f4").tobytes())
    print(f"=> {total} frames decoded and written to {argv[2]}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
