#!/usr/bin/env python3
"""Development bench for the Vorbis (C3) and FLAC (C4) transform kernels: device-resident batches of
BASELINE.json's shapes, events on the launch stream, parity of the first file(s) against the oracle."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "audio-formats_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0


def file_parity(items, want, tolerance=False):
    """Parity block of an end-to-end batch: the first and the last file's delivered floats against the oracle's decode of
    the same bytes (bit-identical, or -- Opus in the default numeric mode -- within one int16 step on < 1 % of the samples)."""
    import oraclelib  # noqa: F401
    want = np.ascontiguousarray(want, np.float32).reshape(-1)
    rec = {"files_checked": 0, "samples": 0, "mismatches": 0, "rms_error": 0.0}
    sq = 0.0
    for it in (items[0], items[-1]):
        got = np.ascontiguousarray(it["pcm"], np.float32).reshape(-1)
        rec["files_checked"] += 1
        if got.size != want.size:
            rec["mismatches"] += max(got.size, want.size)
            continue
        d = got.astype(np.float64) - want
        bits = int((got.view(np.uint32) != want.view(np.uint32)).sum())
        rec["samples"] += int(got.size)
        sq += float((d ** 2).sum())
        if tolerance:
            step = np.abs(d) * 32767.0
            rec["bitwise_mismatches"] = rec.get("bitwise_mismatches", 0) + bits
            rec["int16_flip_rate"] = float((step > 0).mean())
            ok = step.max() <= 1.0001 and (step > 0).mean() < 0.01 and np.sqrt((d ** 2).mean()) <= 1e-5
            rec["mismatches"] += 0 if ok else max(bits, 1)
        else:
            rec["mismatches"] += bits
    rec["rms_error"] = float(np.sqrt(sq / max(rec["samples"], 1)))
    if tolerance:
        rec["mode"] = "tolerance: <= 1 int16 step, < 1 % of the samples, <= 1e-5 RMS"
    return rec


def time_launches(fn, steps, warmup):
    stream = torch.cuda.current_stream()
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for s, e in ev:
        s.record(stream)
        fn()
        e.record(stream)
    torch.cuda.synchronize()
    return [s.elapsed_time(e) for s, e in ev]


def bench_vorbis(dev, files, packets, steps, warmup, seg):
    import afgpu
    import oraclelib
    from afgpu import synthetic
    plan, spec = synthetic.vorbis_batch_device(0x0662, files, packets, dev)
    if seg:
        plan = afgpu.VorbisPlan(plan.packets, plan.channels, plan.bs0, plan.bs1, plan.pflags, seg)
    out = torch.empty(plan.out_floats, dtype=torch.float32, device=dev)
    ms = time_launches(lambda: plan.transform(spec, out), steps, warmup)
    avg = sum(ms) / len(ms) * 1e-3
    # algorithmic bytes: spectrum in + one flag byte per packet + PCM out
    alg = 4 * plan.spec_floats + plan.total_packets + 4 * plan.out_floats
    # parity on the first file
    so, oo = plan.offsets()
    npk = int(plan.packets[0])
    s_end = int(so[npk]) if plan.total_packets > npk else plan.spec_floats
    o_end = int(oo[npk]) if plan.total_packets > npk else plan.out_floats
    want = oraclelib.vorbis_transform(plan.packets[:1], plan.channels[:1], plan.bs0[:1], plan.bs1[:1],
                                      plan.pflags[:npk], so[:npk], oo[:npk], spec[:s_end].cpu().numpy(), o_end)
    got = out[:o_end].cpu().numpy()
    return {"workload": f"{files} x Ogg Vorbis stereo, {packets} packets, blocksize 2048/256",
            "samples_per_step": plan.out_floats, "avg_kernel_ms": avg * 1e3,
            "samples_per_s": plan.out_floats / avg, "achieved_GBs": alg / avg / 1e9,
            "frac": alg / avg / 1e9 / HBM_PEAK_GBS,
            "bitwise_mismatches": int((got.view(np.uint32) != want.view(np.uint32)).sum()),
            "rms_error": float(np.sqrt(np.mean((got.astype(np.float64) - want) ** 2)))}


def bench_flac(dev, files, frames_per_file, steps, warmup, want_float):
    import afgpu
    import oraclelib
    from afgpu import synthetic
    d_frames, d_sub, res, n_frames, total, frames, subframes = synthetic.flac_batch_device(
        0xF1AC, files, frames_per_file, dev)
    out_i = torch.empty(total, dtype=torch.int32, device=dev)
    out_f = torch.empty(total, dtype=torch.float32, device=dev) if want_float else None
    ms = time_launches(lambda: afgpu.flac_transform(n_frames, d_frames, d_sub, res, out_i, out_f), steps, warmup)
    avg = sum(ms) / len(ms) * 1e-3
    alg = 4 * total + 4 * total * (2 if want_float else 1) + n_frames * (32 + 2 * 68)
    nchk = min(n_frames, 2 * frames_per_file)
    cnt = int(frames["in_off"][nchk]) if nchk < n_frames else total
    want = oraclelib.flac_transform(frames[:nchk], subframes[:2 * nchk], res[:cnt].cpu().numpy(), cnt)
    got = out_i[:cnt].cpu().numpy()
    return {"workload": f"{files} x FLAC 16-bit stereo, {frames_per_file} frames of 4096, LPC order 8/12",
            "samples_per_step": total, "avg_kernel_ms": avg * 1e3, "samples_per_s": total / avg,
            "achieved_GBs": alg / avg / 1e9, "frac": alg / avg / 1e9 / HBM_PEAK_GBS,
            "int32_mismatches": int((got != want).sum())}


def bench_qoa(dev, files, seconds, steps, warmup):
    """`files` copies of one encoded stereo file of `seconds` s (the encoder is the slow CPU fixture generator)."""
    import afgpu
    import oraclelib
    n = int(44100 * seconds)
    t = np.arange(n)
    pcm = np.stack([9000 * np.sin(0.02 * t) + 500 * np.random.default_rng(1).standard_normal(n),
                    7000 * np.sin(0.031 * t + 1)], 1).round().astype(np.int16)
    data, _ = oraclelib.qoa_encode(pcm)
    pad = (-data.size) % 8
    one = np.concatenate([data, np.zeros(pad, np.uint8)])
    fr, ch, _, total = afgpu.qoa_frames(data.tobytes())
    frames = np.tile(fr, files)
    k = np.repeat(np.arange(files, dtype=np.uint64), len(fr))
    frames["byte_off"] += k * np.uint64(one.size)
    frames["out_off"] += k * np.uint64(total * ch)
    d_bytes = torch.from_numpy(np.tile(one, files)).to(dev)
    d_frames = torch.from_numpy(frames.view(np.uint8).copy()).to(dev)
    nout = total * ch * files
    d_f = torch.empty(nout, dtype=torch.float32, device=dev)
    ms = time_launches(lambda: afgpu.qoa_transform(len(frames), d_frames, d_bytes, None, d_f), steps, warmup)
    avg = sum(ms) / len(ms) * 1e-3
    want = oraclelib.qoa_transform(fr, data, total * ch)[1]
    got = d_f[:total * ch].cpu().numpy()
    alg = d_bytes.numel() + 4 * nout
    return {"workload": f"{files} x QOA stereo {seconds} s", "samples_per_step": nout, "avg_kernel_ms": avg * 1e3,
            "samples_per_s": nout / avg, "achieved_GBs": alg / avg / 1e9, "frac": alg / avg / 1e9 / HBM_PEAK_GBS,
            "mismatches": int((got.view(np.uint32) != want.view(np.uint32)).sum())}


def bench_celt(dev, streams, frames_per_stream, steps, warmup):
    import afgpu
    import oraclelib
    from afgpu import synthetic
    rb1, recs1, coef1, tot1 = synthetic.celt_batch(0x0905, [frames_per_stream], [2])
    nrec = len(recs1)
    recs = np.tile(recs1, streams)
    k = np.repeat(np.arange(streams, dtype=np.uint64), nrec)
    recs["coef_off"] += k * np.uint64(coef1.size)
    recs["out_off"] += k * np.uint64(tot1)
    rec_base = np.concatenate([(rb1[:-1] + np.uint64(s * nrec)) for s in range(streams)] + [np.array([streams * nrec], np.uint64)])
    d_coef = torch.from_numpy(np.tile(coef1, streams)).to(dev)
    d_recs = torch.from_numpy(recs.view(np.uint8).copy()).to(dev)
    d_rb = torch.from_numpy(rec_base.view(np.int64)).to(dev)
    d_out = torch.empty(tot1 * streams, dtype=torch.float32, device=dev)
    ms = time_launches(lambda: afgpu.celt_transform(len(rec_base) - 1, d_rb, d_recs, d_coef, d_out), steps, warmup)
    avg = sum(ms) / len(ms) * 1e-3
    want = oraclelib.celt_transform(rb1, recs1, coef1, tot1)
    got = d_out[:tot1].cpu().numpy()
    alg = 8 * tot1 * streams
    return {"workload": f"{streams} x Opus/CELT stereo, {frames_per_stream} frames of 960", "samples_per_step": tot1 * streams,
            "avg_kernel_ms": avg * 1e3, "samples_per_s": tot1 * streams / avg, "achieved_GBs": alg / avg / 1e9,
            "frac": alg / avg / 1e9 / HBM_PEAK_GBS, "numeric_mode": ("exact", "tolerance")[afgpu.get_numeric_mode()],
            "bitwise_mismatches": int((got.view(np.uint32) != want.view(np.uint32)).sum()),
            "rms_vs_oracle": float(np.sqrt(np.mean((got.astype(np.float64) - want) ** 2))),
            "rms_signal": float(np.sqrt(np.mean(want.astype(np.float64) ** 2))),
            "int16_flip_rate": float((oraclelib.opus_output(got)[0] != oraclelib.opus_output(want)[0]).mean())}


def bench_flac_e2e(files, frames_per_file, threads):
    """End to end through afg_batch_decode: file bytes in host memory -> host parse (threads) -> H2D -> restore
    kernel -> D2H -> interleaved floats in host memory.  One encoded file replicated `files` times."""
    import time
    import afgpu
    import flac_bitstream as fb
    rng = np.random.default_rng(3)
    n = 4096 * frames_per_file
    t = np.arange(n)
    pcm = np.stack([9000 * np.sin(0.01 * t) + 800 * rng.standard_normal(n),
                    7000 * np.sin(0.013 * t + 1) + 800 * rng.standard_normal(n)], 1).round().astype(np.int64)
    data, _ = fb.encode_file(pcm, 16, 4096, orders=(8, 12), use_fixed_every=1000)
    blobs = [data] * files
    afgpu.batch_decode(blobs[:2], threads)            # warm up (device init, tables)
    job = afgpu.BatchDecoded(blobs, threads)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        job.run()                                     # the C call only: parse + H2D + kernel + D2H
        best = min(best, time.perf_counter() - t0)
    dt = best
    out = [dict(o, pcm=o["pcm"].copy()) if k in (0, files - 1) else dict(o) for k, o in enumerate(job.items)]
    first = out[0]["pcm"][:8].copy()
    job.close()
    t0 = time.perf_counter()
    for _ in range(files):
        afgpu.flac_parse(data)                        # host parse alone, one thread (includes the numpy copies)
    dt_parse = time.perf_counter() - t0
    ok = all(o["status"] == 0 and o["frames"] == n for o in out) and bool(np.isfinite(first).all())
    samples = 2 * n * files
    import oraclelib
    info, frames, subframes, res = afgpu.flac_parse(data)
    want = oraclelib.flac_transform(frames, subframes, res, info["out_samples"], want_float=True)[1]
    return {"workload": f"{files} x FLAC 16-bit stereo, {frames_per_file} frames of 4096 ({len(data)} bytes each)",
            "threads": threads, "all_ok": ok, "parity": file_parity(out, want), "seconds": dt, "samples_per_s_end_to_end": samples / dt,
            "compressed_MBps": len(data) * files / dt / 1e6,
            "host_parse_one_thread_samples_per_s": samples / dt_parse}


def bench_mp3_e2e(files, frames_per_file, threads):
    """End to end through afg_batch_decode for MP3: file bytes -> host parse (sync, Huffman, requantisation, stereo)
    -> H2D -> transform kernel -> D2H -> delivery copy.  One synthetic 128 kbit/s joint-stereo stream replicated."""
    import time
    import afgpu
    import mp3_bitstream as mb
    data, _, cfg = mb.make_file(5, n_frames=frames_per_file, version="mpeg1", sr=0, mode="ms", bitrate_index=9)
    blobs = [data] * files
    afgpu.batch_decode(blobs[:2], threads)
    job = afgpu.BatchDecoded(blobs, threads)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        job.run()
        best = min(best, time.perf_counter() - t0)
    items = [dict(o, pcm=o["pcm"].copy()) if k in (0, files - 1) else dict(o) for k, o in enumerate(job.items)]
    n = items[0]["frames"]
    finite = bool(np.isfinite(items[0]["pcm"]).all())
    job.close()
    ok = all(o["status"] == 0 and o["frames"] == n for o in items) and finite
    samples = 2 * n * files
    import oraclelib
    want = oraclelib.mp3_decode_file(data)["pcm"]
    return {"workload": f"{files} x MP3 128 kbit/s joint stereo, {frames_per_file} frames ({len(data)} bytes each)",
            "threads": threads, "all_ok": ok, "parity": file_parity(items, want), "seconds": best, "samples_per_s_end_to_end": samples / best,
            "compressed_MBps": len(data) * files / best / 1e6}


def bench_qoa_encode(dev, streams, seconds, steps, warmup):
    """QOA encoder kernel (output side): `streams` stereo streams of `seconds` at 44.1 kHz, int16 PCM resident."""
    import afgpu
    import oraclelib
    n = int(seconds * 44100)
    rng = np.random.default_rng(3)
    t = np.arange(n)[:, None]
    one = np.clip(8000 * np.sin(0.02 * (1 + np.arange(2))[None, :] * t) + 2000 * rng.standard_normal((n, 2)), -32768, 32767).astype(np.int16)
    recs, n_in, n_out = afgpu.qoa_encode_layout([(n, 2)] * streams, 44100)
    d_in = torch.from_numpy(np.tile(one.reshape(-1), streams)).to(dev)
    d_recs = torch.from_numpy(recs.view(np.uint8).copy()).to(dev)
    d_out = torch.zeros(n_out, dtype=torch.uint8, device=dev)
    ms = time_launches(lambda: afgpu.qoa_encode(streams, d_recs, d_out, d_pcm_i16=d_in), steps, warmup)
    avg = sum(ms) / len(ms) * 1e-3
    want, _ = oraclelib.qoa_encode(one, 44100)
    got = d_out[:len(want)].cpu().numpy()
    import time
    t0 = time.perf_counter()
    oraclelib.qoa_encode(one, 44100)
    cpu = time.perf_counter() - t0
    return {"workload": f"{streams} x QOA encode, stereo {seconds} s", "samples_per_step": 2 * n * streams,
            "avg_kernel_ms": avg * 1e3, "samples_per_s": 2 * n * streams / avg,
            "oracle_one_thread_samples_per_s": 2 * n / cpu, "byte_mismatches": int((got != want).sum())}


def bench_mixed_e2e(files, threads):
    """One afg_batch_decode call over a mix like BASELINE config C5 (40 % MP3, 25 % Ogg Vorbis, 25 % FLAC, 10 % QOA in
    place of Opus, whose front-end does not exist): file bytes in host memory -> interleaved floats in host memory."""
    import time
    import afgpu
    import flac_bitstream as fb
    import mp3_bitstream as mb
    import oraclelib
    import vorbis_bitstream as vb
    rng = np.random.default_rng(4)
    mp3, _, _ = mb.make_file(5, n_frames=60, version="mpeg1", sr=0, mode="ms", bitrate_index=9)
    ogg = vb.make_file(11, n_packets=64, force_long_only=True, packet_bytes=(200, 600))
    n = 4096 * 16
    t = np.arange(n)
    pcm = np.stack([9000 * np.sin(0.01 * t) + 800 * rng.standard_normal(n), 7000 * np.sin(0.013 * t + 1) + 800 * rng.standard_normal(n)], 1)
    flac, _ = fb.encode_file(pcm.round().astype(np.int64), 16, 4096, orders=(8, 12), use_fixed_every=1000)
    qoa, _ = oraclelib.qoa_encode(pcm[:5120 * 12].round().astype(np.int16), 44100)
    kinds = [("mp3", mp3)] * 8 + [("ogg", ogg)] * 5 + [("flac", flac)] * 5 + [("qoa", qoa.tobytes())] * 2
    blobs = [kinds[i % len(kinds)][1] for i in range(files)]
    afgpu.batch_decode(blobs[:len(kinds)], threads)
    job = afgpu.BatchDecoded(blobs, threads)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        job.run()
        best = min(best, time.perf_counter() - t0)
    items = [dict(o, pcm=o["pcm"].copy()) if k < len(kinds) else dict(o) for k, o in enumerate(job.items)]
    samples = sum(o["frames"] * o["channels"] for o in items)
    ok = all(o["status"] == 0 and o["frames"] > 0 for o in items)
    fi, ff, fs, fr = afgpu.flac_parse(flac)
    qf, qch, _, qtotal = afgpu.qoa_frames(qoa.tobytes())
    wants = {"mp3": oraclelib.mp3_decode_file(mp3)["pcm"], "ogg": oraclelib.vorbis_file_pcm(oraclelib.vorbis_decode_file(ogg)),
             "flac": oraclelib.flac_transform(ff, fs, fr, fi["out_samples"], want_float=True)[1],
             "qoa": oraclelib.qoa_transform(qf, qoa, qtotal * qch)[1]}
    parity = {}
    for k in range(len(kinds)):
        name = kinds[k][0]
        if name not in parity:
            parity[name] = file_parity([items[k]], wants[name])
    per = {}
    for i, o in enumerate(items):
        k = kinds[i % len(kinds)][0]
        per[k] = per.get(k, 0) + o["frames"] * o["channels"]
    job.close()
    return {"workload": f"{files} mixed files in one batch (MP3 {len(mp3)} B, OGG {len(ogg)} B, FLAC {len(flac)} B, QOA {len(qoa)} B)",
            "threads": threads, "all_ok": ok, "parity": parity, "seconds": best, "samples": samples, "samples_by_format": per,
            "samples_per_s_end_to_end": samples / best, "compressed_MBps": sum(len(b) for b in blobs) / best / 1e6}


def bench_vorbis_e2e(files, packets, threads):
    """End to end through afg_batch_decode for Ogg Vorbis: file bytes -> host parse (pages, code books, floor 1,
    residues, coupling) -> H2D -> transform kernel -> D2H.  One synthetic stream (random code books) replicated."""
    import time
    import afgpu
    import vorbis_bitstream as vb
    data = vb.make_file(11, n_packets=packets, force_long_only=True, packet_bytes=(200, 600))
    blobs = [data] * files
    afgpu.batch_decode(blobs[:2], threads)
    job = afgpu.BatchDecoded(blobs, threads)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        job.run()
        best = min(best, time.perf_counter() - t0)
    items = [dict(o, pcm=o["pcm"].copy()) if k in (0, files - 1) else dict(o) for k, o in enumerate(job.items)]
    n = items[0]["frames"]
    ch = items[0]["channels"]
    finite = bool(np.isfinite(items[0]["pcm"]).all())
    job.close()
    ok = all(o["status"] == 0 and o["frames"] == n for o in items) and finite and n > 0
    samples = ch * n * files
    import oraclelib
    want = oraclelib.vorbis_file_pcm(oraclelib.vorbis_decode_file(data))
    return {"workload": f"{files} x Ogg Vorbis stereo 2048/256, {packets} packets ({len(data)} bytes each)",
            "threads": threads, "all_ok": ok, "parity": file_parity(items, want), "seconds": best, "samples_per_s_end_to_end": samples / best,
            "compressed_MBps": len(data) * files / best / 1e6}


def bench_opus_e2e(files, packets, threads):
    """End to end through afg_batch_decode for Ogg Opus (CELT-only): file bytes -> host parse (pages, packet framing, range
    decoder, CELT frame decoder) -> H2D -> transform kernels -> gain / int16 round trip -> D2H.  One generated stream
    (20 ms fullband stereo frames of 160 bytes: 64 kbit/s, random payloads) replicated."""
    import time
    import afgpu
    import opus_bitstream as ob
    rng = np.random.default_rng(12)
    pkts = [ob.packet(rng, 31, True, 0, sizes=[160]) for _ in range(packets)]
    data = ob.ogg_opus(pkts, 2, preskip=312, packets_per_page=50, comments=(b"R128_TRACK_GAIN=-20000",))
    blobs = [data] * files
    afgpu.batch_decode(blobs[:2], threads)
    job = afgpu.BatchDecoded(blobs, threads)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        job.run()
        best = min(best, time.perf_counter() - t0)
    items = [dict(o, pcm=o["pcm"].copy()) if k in (0, files - 1) else dict(o) for k, o in enumerate(job.items)]
    n = items[0]["frames"]
    ch = items[0]["channels"]
    job.close()
    ok = all(o["status"] == 0 and o["frames"] == n for o in items) and n > 0
    samples = ch * n * files
    import oraclelib
    want = oraclelib.opus_file_pcm(oraclelib.opus_decode_file(data))
    tol = afgpu.get_numeric_mode() == afgpu.NUMERIC_TOLERANCE
    return {"workload": f"{files} x Ogg Opus CELT-only stereo, {packets} packets of 20 ms ({len(data)} bytes each; R128 gain -78 dB: programme level)",
            "threads": threads, "all_ok": ok, "parity": file_parity(items, want, tol), "seconds": best, "samples_per_s_end_to_end": samples / best,
            "compressed_MBps": len(data) * files / best / 1e6}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--codec", default="all")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--vorbis-files", type=int, default=1024)
    ap.add_argument("--vorbis-packets", type=int, default=2584)
    ap.add_argument("--vorbis-seg", type=int, default=0)
    ap.add_argument("--flac-files", type=int, default=4096)
    ap.add_argument("--flac-frames", type=int, default=323)
    ap.add_argument("--flac-float", action="store_true")
    ap.add_argument("--e2e-files", type=int, default=2048)
    ap.add_argument("--e2e-threads", type=int, default=0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    res = {}
    if args.codec in ("all", "vorbis"):
        res["vorbis"] = bench_vorbis(dev, args.vorbis_files, args.vorbis_packets, args.steps, args.warmup,
                                     args.vorbis_seg)
        torch.cuda.empty_cache()
    if args.codec in ("all", "flac"):
        res["flac"] = bench_flac(dev, args.flac_files, args.flac_frames, args.steps, args.warmup, args.flac_float)
    if args.codec in ("all", "qoa"):
        torch.cuda.empty_cache()
        res["qoa"] = bench_qoa(dev, 4096, 4.0, args.steps, args.warmup)
    if args.codec == "mp3_e2e":
        res["mp3_e2e"] = bench_mp3_e2e(args.e2e_files, 60, args.e2e_threads)
    if args.codec == "mixed_e2e":
        res["mixed_e2e"] = bench_mixed_e2e(args.e2e_files * 2, args.e2e_threads)
    if args.codec == "qoa_enc":
        res["qoa_enc"] = bench_qoa_encode(dev, 8192, 4.0, args.steps, args.warmup)
    if args.codec == "vorbis_e2e":
        res["vorbis_e2e"] = bench_vorbis_e2e(args.e2e_files, 128, args.e2e_threads)
    if args.codec == "opus_e2e":
        res["opus_e2e"] = bench_opus_e2e(args.e2e_files, 250, args.e2e_threads)
    if args.codec == "flac_e2e":
        res["flac_e2e"] = bench_flac_e2e(args.e2e_files, 8, args.e2e_threads)
    if args.codec in ("all", "celt"):
        torch.cuda.empty_cache()
        res["celt"] = bench_celt(dev, 8192, 200, args.steps, args.warmup)
    if args.codec == "others":
        # what bench.py appends to its line as `other_workloads` (besides the C5 run): the non-headline kernels on dense
        # device-resident batches, and SURVEY 8d's (c) end-to-end rates -- file bytes in host memory -> floats in host memory
        # through afg_batch_decode, PCIe-inclusive -- each with its parity block against the oracle
        res["celt_dense"] = bench_celt(dev, 8192, 200, args.steps, args.warmup)
        torch.cuda.empty_cache()
        res["qoa"] = bench_qoa(dev, 4096, 4.0, args.steps, args.warmup)
        torch.cuda.empty_cache()
        res["mp3_e2e"] = bench_mp3_e2e(args.e2e_files, 60, args.e2e_threads)
        res["vorbis_e2e"] = bench_vorbis_e2e(args.e2e_files, 128, args.e2e_threads)
        res["flac_e2e"] = bench_flac_e2e(args.e2e_files, 8, args.e2e_threads)
        res["opus_e2e"] = bench_opus_e2e(args.e2e_files, 250, args.e2e_threads)
        res["mixed_e2e"] = bench_mixed_e2e(args.e2e_files * 2, args.e2e_threads)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
