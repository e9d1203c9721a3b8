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


# A second stand-in: Chromium's MediaRecorder ENCODES a synthesised signal as Opus in WebM (WebRTC's libopus encoder, round 5).
# Fullband music-like material at these rates makes it choose CELT-only 20 ms frames (TOC configuration 31, three frames to
# a packet): valid streams from an encoder, which the random range-coder input of tests/opus_bitstream.py is not.
FAKE_PLOTLY_ENCODE = r"""
window.Plotly = {
  version: '2.0.0',
  toImage: function (fig, opts) {
    const meta = fig.layout.meta;
    return new Promise(function (resolve) {
      try {
        const ctx = new AudioContext({ sampleRate: 48000 });
        const n = Math.floor(48000 * meta.secs), ch = meta.channels;
        const buf = ctx.createBuffer(ch, n, 48000);
        for (let c = 0; c < ch; c++) {
          const d = buf.getChannelData(c);
          let seed = meta.seed + 77 * c;
          for (let i = 0; i < n; i++) {
            seed = (seed * 1103515245 + 12345) & 0x7fffffff;
            const noise = (seed / 0x7fffffff - 0.5);
            const t = i / 48000;
            if (meta.kind == 0)        // steady tones, a slow tremolo and a noise floor
              d[i] = 0.25 * Math.sin(2 * Math.PI * (220 + 30 * c) * t) + 0.15 * Math.sin(2 * Math.PI * 1760 * t + c) * Math.sin(2 * Math.PI * 3 * t)
                   + 0.1 * Math.sin(2 * Math.PI * 5200 * t) + 0.05 * noise;
            else if (meta.kind == 1)   // a sweep with clicks every 125 ms: transients (short blocks, time-frequency switching)
              d[i] = 0.3 * Math.sin(2 * Math.PI * (300 + 2500 * t) * t + c) + ((i % 6000) < 24 ? 0.6 * noise : 0.0) + 0.02 * noise;
            else                       // decaying plucked notes over a quiet noise bed
              d[i] = 0.4 * Math.exp(-6 * (t % 0.25)) * Math.sin(2 * Math.PI * (330 * (1 + Math.floor(t * 4) % 5 / 4)) * t + c) + 0.01 * noise;
          }
        }
        const dest = ctx.createMediaStreamDestination();
        dest.channelCount = ch;
        const src = ctx.createBufferSource();
        src.buffer = buf; src.connect(dest);
        const rec = new MediaRecorder(dest.stream, { mimeType: 'audio/webm;codecs=opus', audioBitsPerSecond: meta.bps });
        const chunks = [];
        rec.ondataavailable = function (e) { chunks.push(e.data); };
        rec.onstop = function () {
          new Blob(chunks).arrayBuffer().then(function (ab) {
            const u = new Uint8Array(ab); let s = '';
            for (let i = 0; i < u.length; i += 0x8000) s += String.fromCharCode.apply(null, u.subarray(i, i + 0x8000));
            resolve(JSON.stringify({ webm: btoa(s) }));
          });
        };
        src.onended = function () { setTimeout(function () { rec.stop(); }, 200); };
        ctx.resume().then(function () { rec.start(); src.start(); });
        setTimeout(function () { resolve(JSON.stringify({ error: 'timeout', state: ctx.state })); }, (meta.secs + 10) * 1000);
      } catch (e) { resolve(JSON.stringify({ error: String(e) })); }
    });
  }
};
"""


def webm_opus_packets(data):
    """(OpusHead from CodecPrivate, [packet bytes]) of a WebM file with one Opus track: the SimpleBlocks / Blocks, no lacing."""
    def vint(b, p, keep=False):
        first, n, mask = b[p], 1, 0x80
        while n <= 8 and not (first & mask):
            n += 1
            mask >>= 1
        v = first if keep else first & (mask - 1)
        for i in range(1, n):
            v = (v << 8) | b[p + i]
        return v, p + n, n

    out = {"head": None, "blocks": []}

    def walk(p, end):
        while p < end:
            ident, p2, _ = vint(data, p, True)
            size, p3, n = vint(data, p2)
            e = end if size == (1 << (7 * n)) - 1 else p3 + size            # unknown size: to the end of the parent
            if ident in (0x18538067, 0x1654AE6B, 0xAE, 0x1F43B675, 0xA0):   # Segment, Tracks, TrackEntry, Cluster, BlockGroup
                walk(p3, min(e, end))
            elif ident == 0x63A2:
                out["head"] = bytes(data[p3:e])
            elif ident in (0xA3, 0xA1):                                       # SimpleBlock / Block: track, int16 timecode, flags
                _, q, _ = vint(data, p3)
                assert data[q + 2] & 0x06 == 0, "laced block"
                out["blocks"].append(bytes(data[q + 3:e]))
            p = e
    walk(0, len(data))
    return out["head"], out["blocks"]


class WebEncoder:
    def __init__(self):
        from kaleido.scopes.plotly import PlotlyScope
        self.dir = tempfile.mkdtemp()
        js = os.path.join(self.dir, "fake_plotly_encode.js")
        with open(js, "w") as fh:
            fh.write(FAKE_PLOTLY_ENCODE)
        self.scope = PlotlyScope(plotlyjs=Path(js).as_uri())
        self.scope.chromium_args += ("--autoplay-policy=no-user-gesture-required",)

    def encode_ogg_opus(self, kind, channels, secs, bps, seed):
        """-> an Ogg Opus file of CELT-only packets made by Chromium's encoder (remuxed from its WebM), or None"""
        import opus_bitstream as ob
        out = self.scope.transform({"data": [], "layout": {"meta": {"kind": kind, "channels": channels, "secs": secs, "bps": bps, "seed": seed}}},
                                   format="json")
        d = json.loads(out.decode() if isinstance(out, bytes) else out)
        if "webm" not in d:
            raise RuntimeError(str(d))
        head, pkts = webm_opus_packets(base64.b64decode(d["webm"]))
        if head is None or not pkts or any((p[0] >> 3) < 16 for p in pkts):   # a SILK or hybrid packet: not this path
            return None
        return ob.ogg_opus(pkts, head[9], preskip=head[10] | (head[11] << 8), gain=0, head=head, trim=0)


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


def generated_streams(wa, out):
    """Round 5: FFmpeg's mp3float and vorbis decoders on files from this repository's own bitstream writers -- random code
    words, but legal streams: every Huffman table, linbits escapes, scfsi, all block types in encoder order, MPEG-1/2/2.5;
    every code-book kind, residue types 1 and 2, multi-stage cascades, channel coupling.  Left out on purpose, each because the
    two decoder families are KNOWN to read it differently (tests/test_oracle_independent.py has the details): MP3 mixed
    blocks and intensity stereo (the writer's right channel is not empty above the intensity bound, as an encoder's is); Vorbis
    residue type 0, sequence_p books, lookup type 2 (FFmpeg refuses it), packets that end early."""
    import mp3_bitstream as mb
    import vorbis_bitstream as vb
    import oraclelib
    mb.MIXED_P = 0.0
    k = 0
    for seed, (ver, sr, mode) in enumerate([("mpeg1", 0, "stereo"), ("mpeg1", 1, "ms"), ("mpeg1", 2, "mono"), ("mpeg2", 0, "ms"),
                                            ("mpeg2", 2, "stereo"), ("mpeg25", 1, "ms")]):
        data, _, cfg = mb.make_file(600 + seed, n_frames=10, version=ver, sr=sr, mode=mode, strict=True)
        out[f"gen_mp3_{k}_file"] = np.frombuffer(data, np.uint8)
        out[f"gen_mp3_{k}_pcm"] = wa.decode(data, 1 if mode == "mono" else 2, cfg["hz"])
        out[f"gen_mp3_{k}_rate"] = np.array([cfg["hz"]])
        k += 1
    mb.MIXED_P = 0.4
    vb.LOOKUP1_ONLY = True
    vb.SIMPLE = {"no_seq", "one_submap", "plain_coupling", "no_short_packets"}
    kept = tried = decoded = 0
    seed = 820
    while kept < 6 and tried < 40:
        tried += 1
        seed += 1
        try:
            data = vb.make_file(seed, n_packets=6, packet_bytes=(9000, 10000), residue_types=(1, 2))
        except ValueError:
            continue
        rec = oraclelib.vorbis_decode_file(data)
        got = oraclelib.vorbis_file_pcm(rec)
        if got.size == 0 or not np.abs(got).max() > 0:
            continue
        try:
            ref = wa.decode(data, got.shape[1], 44100)
        except RuntimeError:
            continue
        decoded += 1
        n = min(len(got), len(ref))
        d = got[:n].astype(np.float64) - ref[:n]
        if len(ref) < len(got) or np.sqrt(np.mean(d ** 2)) > 1e-4 * np.sqrt(np.mean(ref[:n].astype(np.float64) ** 2)):
            continue                                    # FFmpeg took one of its error paths on this random content: not comparable
        out[f"gen_ogg_{kept}_file"] = np.frombuffer(data, np.uint8)
        out[f"gen_ogg_{kept}_pcm"] = ref
        kept += 1
    out["gen_ogg_selection"] = np.array([tried, decoded, kept])
    vb.LOOKUP1_ONLY = False
    vb.SIMPLE = set()


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
    # encoder-made CELT: kept only if every packet is CELT-only (TOC configuration >= 16)
    we = WebEncoder()
    for k, (kind, ch, secs, bps) in enumerate(((0, 2, 1.5, 256000), (1, 2, 1.5, 160000), (2, 1, 1.5, 128000))):
        ogg = we.encode_ogg_opus(kind, ch, secs, bps, 1000 + k)
        if ogg is None:
            print("encoder-made file", k, "holds SILK / hybrid packets: left out")
            continue
        out[f"opus_enc{k}_file"] = np.frombuffer(ogg, np.uint8)
        out[f"opus_enc{k}_pcm"] = wa.decode(ogg, ch, 48000)
    generated_streams(wa, out)
    np.savez_compressed(os.path.join(HERE, "independent_webaudio.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, v.dtype)


if __name__ == "__main__":
    main()
