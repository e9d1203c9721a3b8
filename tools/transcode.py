#!/usr/bin/env python3
"""transcode [--format s24|s16|s8|f32] [--no-dither | --dither-seed N] input.{mp3|flac|ogg|qoa} output.{wav|qoa}

The reference's examples/transcode flow (main.d:12-84) over the device library: open, report format / rate /
channels / length, read 1024-frame chunks, write them out with the library's own writers (afg_wav_encode[_dithered] /
afg_qoa_encode_hip).  Defaults are the example's: 24-bit PCM with TPDF dither drawn from libc rand() (main.d:54-56).
--dither-seed N draws from a seeded LCG instead so that the bytes are reproducible (tests/test_transcode_gpu.py);
--format f32 writes the decoded floats untouched (BASELINE config C1 is checked on those)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-formats_amd"))

import numpy as np  # noqa: E402


def lcg(seed):
    """The generator --dither-seed feeds the dither: 31-bit LCG (the classic rand() constants)."""
    state = [seed & 0x7fffffff]

    def draw():
        state[0] = (state[0] * 1103515245 + 12345) & 0x7fffffff
        return state[0]
    return draw


def main(argv):
    import argparse
    ap = argparse.ArgumentParser(prog="transcode", usage="transcode [options] input.{mp3|flac|ogg|qoa} output.{wav|qoa}")
    ap.add_argument("--format", default="s24", choices=["s24", "s16", "s8", "f32"])
    ap.add_argument("--no-dither", action="store_true")
    ap.add_argument("--dither-seed", type=int, default=None)
    ap.add_argument("input")
    ap.add_argument("output")
    try:
        args = ap.parse_args(argv[1:])
    except SystemExit:
        return 2
    argv = [argv[0], args.input, args.output]
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
    else:                                                # the library's WAV writer
        fmt = {"s24": afgpu.WAV_S24LE, "s16": afgpu.WAV_S16LE, "s8": afgpu.WAV_S8, "f32": afgpu.WAV_FP32LE}[args.format]
        dither = None if (args.no_dither or args.format == "f32") else ("libc" if args.dither_seed is None else lcg(args.dither_seed))
        out = afgpu.wav_encode(pcm, int(rate), fmt, dither=dither)
    with open(argv[2], "wb") as fh:
        fh.write(out)
    print(f"=> {total} frames decoded and written to {argv[2]}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
