"""development soak (GPU): randomly damaged QOA / FLAC / Ogg Vorbis / MP3 / Ogg Opus files through afg_batch_decode against the oracle's
decode of the same damaged bytes FROM THE BYTES (every codec through the oracle's own front-end: oracle/flac_frontend.c and the QOA
stream layer of oracle/qoa_lms.c since round 5) -- statuses, lengths and samples (QOA / FLAC bit for bit, the float codecs within
1e-5 RMS absolute).  Counted, not skipped: files the product refuses although the oracle decodes them, and FLAC files where the
product stops at a frame the reference delivers from a stale decode buffer (oracle flag AFGO_FLAC_F_IGNORED_FAILURE).
usage: python tools/soak_damaged.py [rounds]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-formats_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import afgpu
import oraclelib
import flac_bitstream as fb
import vorbis_bitstream as vb
import mp3_bitstream as mb
import opus_bitstream as ob
from test_flac_frontend import make_pcm
from test_stream_gpu import qoa_file, read_all

rng = np.random.default_rng(2024)


def damage(data, lo):
    d = bytearray(data)
    for _ in range(int(rng.integers(1, 6))):
        p = int(rng.integers(lo, len(d)))
        if rng.random() < 0.5:
            d[p] ^= 1 << int(rng.integers(0, 8))
        else:
            d[p] = int(rng.integers(0, 256))
    if rng.random() < float(os.environ.get("AFG_SOAK_CUT", "0.2")):          # cut short (AFG_SOAK_CUT: how often)
        d = d[:int(rng.integers(max(lo, len(d) // 2), len(d)))]
    return bytes(d)


def want_qoa(data):
    o = oraclelib.qoa_decode_file(data)
    if isinstance(o, int) or len(o["pcm"]) == 0:
        return None
    return (o["pcm"].astype(np.float32) * np.float32(1.0 / 32767)).reshape(-1, o["channels"])      # qoa.d:831-838


FLAC_STALE = {"n": 0}


def want_flac(data):
    """oracle front-end; a frame the reference delivers after an ignored subframe failure (stale decode buffer,
    drflac.d:1591-1594) is where the product ends the stream: expected = everything before it"""
    o = oraclelib.flac_decode_file(data)
    if isinstance(o, int):
        return None
    pcm = o["pcm"]
    if o["flags"] and o["first_flag_sample"] is not None:
        pcm = pcm[:o["first_flag_sample"]]
        FLAC_STALE["n"] += 1
    return (pcm.astype(np.float64) * (1.0 / 2147483647.0)).astype(np.float32).reshape(-1, o["channels"])   # stream.d:505-511


OGG_WINDOW_CUT = {"n": 0, "eof": 0, "phantom": 0}
OGG_PHANTOM = {}          # file length -> the delivered frames without upstream's phantom last packet


def want_ogg(data):
    """oracle front-end + transform; a packet whose left window does not have the length of the previous packet's right
    window (damaged flags: the reference mixes the two lengths, stb_vorbis2.d:2618-2627) is where the product ends the
    stream (HISTORY.md 4): expected = everything before it"""
    rec = oraclelib.vorbis_decode_file(data)
    if rec is None:
        return None
    if rec["pcm_frames"] == 0:
        # the reference's stream-length scan at open ran into the end of the data (no intact last page) and its seek back
        # leaves the eof flag set: nothing is delivered.  The product decodes what is there, as upstream stb_vorbis does
        # (tests/test_vorbis_frontend.py pins both): such a file is held against the oracle with upstream's seek
        rec = oraclelib.vorbis_decode_file(data, seek_clears_eof=True)
        if rec is None:
            return None
        OGG_WINDOW_CUT["eof"] += 1
        # Upstream's reader then makes one more packet out of a page header cut short by the end of the data: the header
        # fields read as zeros, "no segments", and the next packet takes the PREVIOUS page's first lacing value and reads
        # zeros past the end -- a short block of silence.  The product stops at the cut header: either length is accepted
        # for such a file (counted), the samples before it must agree.
        # (likewise a packet that continues onto a page that is not there: upstream decodes the part it has, the product ends
        #  the stream in front of it)
        n = len(rec["pflags"])
        if n and not rec["pflags"][-1] & 1 and rec["take_count"][-1] > 0:
            OGG_PHANTOM[len(data)] = int(rec["pcm_frames"]) - int(rec["take_count"][-1])
    pcm = oraclelib.vorbis_file_pcm(rec)
    bs0, bs1, prev = rec["blocksize0"], rec["blocksize1"], 0
    for q, fl in enumerate(rec["pflags"]):
        lng = bool(fl & 1)
        n = bs1 if lng else bs0
        left = (n + bs0) // 4 - (n - bs0) // 4 if (lng and not fl & 2) else n // 2
        right = bs0 // 2 if (lng and not fl & 4) else n // 2
        if prev and prev != left:
            OGG_WINDOW_CUT["n"] += 1
            return pcm[:int(np.sum(rec["take_count"][:q]))]
        prev = right
    return pcm


def want_mp3(data):
    rec = oraclelib.mp3_decode_file(data)
    return None if rec is None else rec["pcm"].reshape(-1, rec["channels"])


def want_opus(data):
    rec = oraclelib.opus_decode_file(data)
    return None if isinstance(rec, int) or rec.get("error") else oraclelib.opus_file_pcm(rec)


def dump_bad(tag, r, kind, d):
    """AFG_SOAK_DUMP=<dir>: keep the bytes of a file the soak flags"""
    where = os.environ.get("AFG_SOAK_DUMP")
    if where:
        with open(os.path.join(where, f"{tag}_{r}_{kind}_{len(d)}.bin"), "wb") as fh:
            fh.write(d)


WIDE = bool(os.environ.get("AFG_SOAK_WIDE"))
HDR = bool(os.environ.get("AFG_SOAK_HDR"))          # the damage may land in the headers too


def run(rounds, seed=2024, streams=True):
    """-> (decoded, rejected, disagreements); run.refused / run.flac_stale / run.ogg_window_cut: the counted divergence classes"""
    global rng
    rng = np.random.default_rng(seed)
    bad = 0
    n_ok = n_rejected = 0
    run.refused = {"ogg": 0, "opus": 0}
    run.over_full_scale = 0
    FLAC_STALE["n"] = 0
    OGG_WINDOW_CUT["n"] = OGG_WINDOW_CUT["eof"] = OGG_WINDOW_CUT["phantom"] = 0
    OGG_PHANTOM.clear()
    for r in range(rounds):
        files, wants, kinds = [], [], []
        for k in range(15):
            kind = ("qoa", "flac", "ogg", "mp3", "opus")[k % 5]
            if kind == "qoa":
                base, _ = qoa_file(int(rng.integers(3000, 30000)), int(rng.integers(1, 3)), 44100, int(rng.integers(0, 1 << 30)))
                d = damage(base, 8)
                w = want_qoa(d)
            elif kind == "flac":
                if WIDE:                                             # AFG_SOAK_WIDE=1: other sample widths, block sizes, channel counts
                    bps = (16, 24, 8, 12, 20)[int(rng.integers(0, 5))]
                    base, _ = fb.encode_file(make_pcm(int(rng.integers(1500, 12000)), int(rng.integers(1, 4)), bps, int(rng.integers(0, 1 << 30))), bps,
                                             (4096, 1152, 576, 2304, 4608)[int(rng.integers(0, 5))], orders=(8, 12, 2, 32)[:int(rng.integers(2, 5))],
                                             use_fixed_every=(1000, 3)[int(rng.integers(0, 2))])
                else:
                    base, _ = fb.encode_file(make_pcm(int(rng.integers(2000, 20000)), 2, 16, int(rng.integers(0, 1 << 30))), 16, 4096, orders=(8, 12, 2))
                d = damage(base, 4 if HDR else 42 if not WIDE else (42, 8)[int(rng.integers(0, 2))])
                w = want_flac(d)
            elif kind == "mp3":
                if WIDE:                                             # MPEG-2 / 2.5 rates, intensity stereo, free choice of the sampling rate
                    # (not MPEG-2.5 at 8 kHz: its mixed blocks make L3_reorder read the reference's uninitialised stack scratch,
                    #  minimp3.d:1218-1223 -- what comes out depends on what the thread decoded before, in the oracle as in the product)
                    version = ("mpeg1", "mpeg2", "mpeg25")[int(rng.integers(0, 3))]
                    base = mb.make_file(int(rng.integers(0, 1 << 20)), n_frames=int(rng.integers(8, 30)), version=version, sr=int(rng.integers(0, 2 if version == "mpeg25" else 3)),
                                        mode=("stereo", "ms", "mono", "intensity", "ms+intensity")[int(rng.integers(0, 5))])[0]
                else:
                    base = mb.make_file(int(rng.integers(0, 1 << 20)), n_frames=int(rng.integers(8, 40)), mode=("stereo", "ms", "mono")[int(rng.integers(0, 3))])[0]
                d = damage(base, 4)
                w = want_mp3(d)
            elif kind == "opus":
                base = ob.random_celt_file(rng, int(rng.integers(1, 3)), int(rng.integers(10, 40)), pcm_rms=0.05)[0]
                d = damage(base, 28 if HDR else len(base) // 2)
                w = want_opus(d)
            else:
                base = None
                while base is None:                                  # (the writer's random set-up has dead ends for about one seed in 250)
                    try:
                        # every stream shape the tolerance-mode walk has a kernel for, and the general kernel's
                        ch, bs = [(2, (256, 2048)), (1, (256, 2048)), (2, (256, 1024)), (1, (512, 1024)), (2, (512, 4096)), (1, (256, 4096)),
                                  (3, (256, 2048)), (6, (256, 1024)), (2, (1024, 2048)), (2, (256, 2048))][int(rng.integers(0, 10))]
                        base = vb.make_file(int(rng.integers(0, 1 << 20)), channels=ch, bs=bs, n_packets=30, force_long_only=bool(rng.integers(0, 2)))
                    except ValueError:
                        pass
                d = damage(base, 60 if HDR else len(base) // 3)
                w = want_ogg(d)
            files.append(d); wants.append(w); kinds.append(kind)
        res = afgpu.batch_decode(files, n_threads=4)
        for kind, out, w, d in zip(kinds, res, wants, files):
            if w is None or len(w) == 0:
                n_rejected += 1
                if out["status"] == 0 and out["frames"] > 0 and kind in ("qoa", "flac"):
                    print("product decoded what the oracle rejects", kind, out["frames"]); bad += 1
                continue
            if out["status"] != 0:
                if kind in ("ogg", "opus"):                          # (the product ends a stream at an inconsistent window, HISTORY.md 4): counted
                    run.refused[kind] += 1
                    continue
                print("product rejected", kind, out["message"]); bad += 1; continue
            n_ok += 1
            got = out["pcm"] if out["pcm"] is not None else np.zeros((0, max(1, out["channels"])), np.float32)
            if streams and out["frames"] > 0:
                # the same bytes pulled through the AudioStream surface in odd-sized reads: chunked decoding must deliver the
                # batch path's samples bit for bit
                st = afgpu.AudioStream()
                st.openFromMemory(d)
                if st.isError():
                    print("stream refuses what the batch decodes", kind, st.errorMessage()); bad += 1
                else:
                    pulled = read_all(st, out["channels"], int(rng.integers(300, 5000)))
                    m = min(len(pulled), len(got))
                    # (a stream that hits damage reports the error after delivering what came before it)
                    if (len(pulled) != len(got) and not st.isError()) or not np.array_equal(pulled[:m].view(np.uint32), got[:m].view(np.uint32)):
                        print("stream != batch", kind, len(pulled), len(got), st.isError()); bad += 1
                        if os.environ.get("AFG_SOAK_DUMP"):
                            with open(os.path.join(os.environ["AFG_SOAK_DUMP"], f"stream_ne_batch_{r}_{kind}_{len(d)}.bin"), "wb") as fh:
                                fh.write(d)
                            for i, blob in enumerate(files):
                                with open(os.path.join(os.environ["AFG_SOAK_DUMP"], f"round_{r}_file_{i:02d}_{kinds[i]}.bin"), "wb") as fh:
                                    fh.write(blob)
                st.cleanUp()
            n = min(len(got), len(w))
            diff = got[:n].astype(np.float64) - w[:n]
            if kind == "opus":
                # the decoder's int16 / 32767: the same values, or (default mode) a rare neighbour
                far = np.abs(diff) > 1 / 32767 + 1.2e-7
                # Damage that lands in the band energies can put the decoder's floats 1e9 times over full scale (the int16
                # conversion clips them): float32 carries 1.2e-7 of THAT, so where the clipped waveform passes through the
                # int16 range a default-mode sample may land anywhere.  Such a file (the oracle's output sits at the rails
                # for more than 0.5 % of its samples; counted) is held to 0.1 % of its samples; exact mode stays bit-exact.
                railed = n and float(np.mean(np.abs(w[:n]) >= 32766.5 / 32767)) > 0.005
                if railed:
                    run.over_full_scale += 1
                if len(got) != len(w) or (n and far.any() and not (railed and far.mean() <= 0.001 and afgpu.get_numeric_mode() == afgpu.NUMERIC_TOLERANCE)):
                    print("opus mismatch", len(got), len(w), float(np.abs(diff).max()) if n else None); bad += 1
                    dump_bad("mismatch", r, kind, d)
            elif kind in ("ogg", "mp3"):
                if len(got) != len(w):
                    if kind == "ogg" and OGG_PHANTOM.get(len(d)) == len(got):
                        OGG_WINDOW_CUT["phantom"] += 1
                    else:
                        print(kind, "length", len(got), len(w)); bad += 1
                        dump_bad("length", r, kind, d)
                rms = float(np.sqrt(np.mean(diff ** 2))) if n else 0.0
                sig = float(np.sqrt(np.mean(w[:n].astype(np.float64) ** 2))) if n else 0.0
                # absolute 1e-5 of full scale: the generators keep the undamaged signal inside it (round 5).  Damage that lands in
                # a gain field can lift the decode to hundreds of times full scale; float32 carries 1.2e-7 of THAT, so such a
                # file (counted: run.over_full_scale) is held to 1e-5 of its own level
                if sig > 1.0:
                    run.over_full_scale += 1
                if rms > 1e-5 * max(1.0, sig):
                    print(kind, "mismatch", len(got), len(w), rms, "signal rms", sig); bad += 1
                    dump_bad("mismatch", r, kind, d)
            else:
                if len(got) != len(w) or not np.array_equal(got.view(np.uint32), w.astype(np.float32).view(np.uint32)):
                    print(kind, "mismatch", len(got), len(w)); bad += 1
                    dump_bad("mismatch", r, kind, d)
    run.flac_stale = FLAC_STALE["n"]
    run.ogg_window_cut = OGG_WINDOW_CUT["n"]
    run.ogg_eof_quirk = OGG_WINDOW_CUT["eof"]
    return n_ok, n_rejected, bad


if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    n_ok, n_rejected, bad = run(rounds, int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
    print("rounds", rounds, "decoded", n_ok, "rejected", n_rejected, "bad", bad, "| product refused what the oracle decodes:", run.refused,
          "| FLAC streams ended at a stale-buffer frame:", run.flac_stale, "| Ogg streams ended at an inconsistent window:", run.ogg_window_cut, "| Ogg files the reference stops after its length scan (held against upstream's seek):", run.ogg_eof_quirk,
          "of which end one phantom packet early:", OGG_WINDOW_CUT["phantom"], "| damaged MP3 / Ogg / Opus files decoding over full scale:", run.over_full_scale)
    sys.exit(1 if bad else 0)
