#!/usr/bin/env python3
"""tests/golden/make_independent.py -- PCM of the fixture files from a decoder NOT written in this repository.

The reference is D (no compiler here) and ships no vectors, so every other fixture under tests/golden/ comes from this
repository's own oracle.  This script produces the one exception: the image's `kaleido` package embeds a headless
Chromium whose WebAudio `decodeAudioData` runs Chromium's FFmpeg build (mp3float, vorbis, flac) and libopus.  kaleido
loads whatever script it is given as "plotly.js"; the stand-in below implements `Plotly.toImage` as "decode the audio
bytes in layout.meta and return the PCM", and kaleido's JSON export hands the string back.

Runs in the BUILD container only (kaleido is not part of the product or of the GPU box's test run); the vectors it
writes -- tests/golden/independent_webaudio.npz -- are data: decoder outputs plus the generated input files.
It is not the reference, so it does not pin the oracle in the sense of SURVEY 8c; tests/test_oracle_independent.py
states what agreement it does show.
"""
import base64
import json
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "audio-formats_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

FAKE_PLOTLY = r"""
window.Plotly = {
  version: '2.0.0',
  toImage: function (fig, opts) {
    const meta = fig.layout.meta;
    const bin = atob(meta.b64);
    const bytes = new Uint8Array(bin.length);
    for (let i = 0; i < bin.length; i++) bytes[i] = bin.charCodeAt(i);
    const ctx = new OfflineAudioContext(meta.channels, 1, meta.rate);        // the file's own rate: no resampling
    return ctx.decodeAudioData(bytes.buffer).then(function (buf) {
      const out = { rate: buf.sampleRate, length: buf.length, channels: buf.numberOfChannels, pcm: [] };
      for (let c = 0; c < buf.numberOfChannels; c++) {
        const f = buf.getChannelData(c);
        const u = new Uint8Array(f.buffer, f.byteOffset, f.byteLength);
        let s = '';
        for (let i = 0; i < u.length; i += 0x8000) s += String.fromCharCode.apply(null, u.subarray(i, i + 0x8000));
        out.pcm.push(btoa(s));
      }
      return JSON.stringify(out);
    }, function (err) { return JSON.stringify({ error: String(err) }); });
  }
};
"""


class WebAudio:
    def __init__(self):
        from kaleido.scopes.plotly import PlotlyScope
        self.dir = tempfile.mkdtemp()
        js = os.path.join(self.dir, "fake_plotly.js")
        with open(js, "w") as fh:
            fh.write(FAKE_PLOTLY)
        self.scope = PlotlyScope(plotlyjs=Path(js).as_uri())

    def decode(self, data, channels, rate):
        fig = {"data": [], "layout": {"meta": {"b64": base64.b64encode(data).decode(), "channels": channels, "rate": rate}}}
        out = self.scope.transform(fig, format="json")
        d = json.loads(out.decode() if isinstance(out, bytes) else out)
        if "error" in d:
            raise RuntimeError(d["error"])
        assert d["rate"] == rate
        return np.stack([np.frombuffer(base64.b64decode(c), np.float32) for c in d["pcm"]], 1)


def opus_pair(seed, channels, n_packets):
    """The same random CELT packets (20 ms fullband frames) muxed twice: with header gain 0 (what the oracle decodes: the
    reference reads the header gain unsigned, so it cannot be given a negative one) and with -78.125 dB (what libopus
    applies: the random payloads decode ~70 dB over full scale, and WebAudio clips at +-1)."""
    import opus_bitstream as ob
    rng = np.random.default_rng(seed)
    _, pkts = ob.random_celt_file(rng, channels, n_packets, preskip=312, mixed_stereo=False, configs=[31], codes=[0])
    return (ob.ogg_opus(pkts, channels, 312, 0, (), trim=0), ob.ogg_opus(pkts, channels, 312, (-20000) & 0xffff, (), trim=0))


def main():
    import flac_bitstream as fb
    from test_flac_frontend import make_pcm
    wa = WebAudio()
    out = {}
    for name, ext in (("mp3", "mp3"), ("ogg", "ogg")):
        data = open(os.path.join(HERE, "mathjax_invalid_keypress." + ext), "rb").read()
        out[name + "_pcm"] = wa.decode(data, 2, 44100)
    pcm = make_pcm(20000, 2, 16, 5)
    flac, _ = fb.encode_file(pcm, 16, 4096, orders=(8, 12))
    out["flac_file"] = np.frombuffer(flac, np.uint8)
    out["flac_source"] = pcm.astype(np.int32)
    out["flac_pcm"] = wa.decode(flac, 2, 44100)
    for tag, seed, ch in (("opus_stereo", 11, 2), ("opus_mono", 16, 1)):
        a, b = opus_pair(seed, ch, 60)
        out[tag + "_file_gain0"] = np.frombuffer(a, np.uint8)
        out[tag + "_file_gain_m78dB"] = np.frombuffer(b, np.uint8)
        out[tag + "_pcm_m78dB"] = wa.decode(b, ch, 48000)
    np.savez_compressed(os.path.join(HERE, "independent_webaudio.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, v.dtype)


if __name__ == "__main__":
    main()
