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
    """Parity block of an end-to-end batch: the given items' delivered floats against the oracle's decode of the same bytes.
    tolerance False: bit-identical; "rms": <= 1e-5 RMS (Vorbis in the default numeric mode: a different factorisation of the
    inverse MDCT); True: Opus in the default numeric mode -- within one int16 step on < 1 % of the samples and <= 1e-5 RMS."""
    import oraclelib  # noqa: F401
    want = np.ascontiguousarray(want, np.float32).reshape(-1)
    rec = {"files_checked": 0, "samples": 0, "mismatches": 0, "rms_error": 0.0}
    sq = 0.0
    for it in items:
        got = np.ascontiguousarray(it["pcm"], np.float32).reshape(-1)
        rec["files_checked"] += 1
        if got.size != want.size:
            rec["mismatches"] += max(got.size, want.size)
            continue
        d = got.astype(np.float64) - want
        bits = int((got.view(np.uint32) != want.view(np.uint32)).sum())
        rec["samples"] += int(got.size)
        sq += float((d ** 2).sum())
        sig = float(np.sqrt((want.astype(np.float64) ** 2).mean())) if want.size else 0.0
        rec.setdefault("rms_signal_per_file", []).append(sig)
        if sig == 0.0:
            rec["silent_files"] = rec.get("silent_files", 0) + 1          # a file that decodes to silence checks nothing: counted as a failure
            rec["mismatches"] += 1
        if tolerance:
            rec["bitwise_mismatches"] = rec.get("bitwise_mismatches", 0) + bits
            # ABSOLUTE: the generated files decode to an encoder's level (rms about 0.05 of full scale, tools/e2e_files.py), so
            # 1e-5 rms of full scale is what north_star's tolerance says
            ok = np.sqrt((d ** 2).mean()) <= 1e-5 and not np.isnan(got).any()
            if tolerance is True:
                step = np.abs(d) * 32767.0
                rec["int16_flip_rate"] = float((step > 0).mean())
                ok = ok and step.max() <= 1.004 and (step > 0).mean() < 0.01      # (int16 / 32767 near full scale: neighbours are 1 / 32767 +- 1.2e-7 apart)
            rec["mismatches"] += 0 if ok else max(bits, 1)
        else:
            rec["mismatches"] += bits
    rec["rms_error"] = float(np.sqrt(sq / max(rec["samples"], 1)))
    if tolerance is True:
        rec["mode"] = "tolerance: <= 1 int16 step, < 1 % of the samples, <= 1e-5 RMS"
    elif tolerance:
        rec["mode"] = "tolerance: <= 1e-5 RMS of full scale, absolute"
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


# ----------------------------------------------------------------------------------------------------------------
# end to end, file bytes in host memory -> floats in host memory (PCIe-inclusive, never bench.py's `value`)
# ----------------------------------------------------------------------------------------------------------------
from e2e_files import E2E_DISTINCT  # noqa: E402  distinct generated files per codec; a batch repeats them (as distinct buffers) up to --e2e-files
E2E_PASSES = 5              # the rate is the median over this many timing windows ...
E2E_WINDOW_S = 1.0          # ... each at least this long (back-to-back afg_batch_decode calls over the whole batch)


from e2e_files import generate_files  # noqa: E402  (tools/e2e_files.py: the generators, run in worker processes)


def batch_of(distinct, files):
    """`files` blobs cycling through the distinct ones, every one its own buffer."""
    return [bytes(bytearray(distinct[i % len(distinct)])) for i in range(files)]


def timed_batch(blobs, threads):
    """afg_batch_decode over the whole batch, back to back: median seconds per call over E2E_PASSES windows of >= E2E_WINDOW_S, and
    the CPU seconds (all threads of the process) a call consumes -- the host parse is the part of a decode that stays on the
    host, and on a box with a CPU quota it bounds the end-to-end rate whatever the pipeline does."""
    import time
    import afgpu
    afgpu.batch_decode(blobs[:2], threads)            # warm up (device init, tables)
    job = afgpu.BatchDecoded(blobs, threads)
    job.run()
    per_call, cpu_per_call = [], []
    for _ in range(E2E_PASSES):
        n, t0, c0 = 0, time.perf_counter(), time.process_time()
        while True:
            job.run()                                 # the C call only: parse + H2D + kernels + D2H
            n += 1
            dt = time.perf_counter() - t0
            if dt >= E2E_WINDOW_S:
                break
        per_call.append(dt / n)
        cpu_per_call.append((time.process_time() - c0) / n)
    order = sorted(range(len(per_call)), key=lambda i: per_call[i])
    mid = order[len(order) // 2]
    timed_batch.last_cpu_seconds_per_call = cpu_per_call[mid]
    per_call.sort()
    return job, per_call[len(per_call) // 2], per_call


def host_cpu_bound(samples, sec):
    """What the host side of a call costs: CPU seconds per call (process-wide, every helper thread), the CPUs that keeps busy,
    and -- where the container has a CPU quota -- the rate at which the quota alone would cap back-to-back calls."""
    sys.path.insert(0, ROOT)
    from bench import host_cpu_info
    cpus, quota = host_cpu_info()
    cpu_s = getattr(timed_batch, "last_cpu_seconds_per_call", None)
    if not cpu_s:
        return {}
    out = {"host_cpu_seconds_per_call": cpu_s, "host_cpus_busy": cpu_s / sec, "cgroup_cpu_quota": quota}
    if quota:
        out["samples_per_s_at_the_cpu_quota"] = samples / (cpu_s / quota)
        out["host_cpu_note"] = ("bitstream parsing stays on the host (north_star): a call needs this many CPU seconds whatever the device does; "
                                "with the quota's CPUs that alone allows `samples_per_s_at_the_cpu_quota`")
    return out


_CPU_E2E_CACHE = {}


def cpu_e2e(kind, distinct, threads_hint=0):
    key = (kind, len(distinct), hash(distinct[0]), hash(distinct[-1]))
    if key not in _CPU_E2E_CACHE:
        _CPU_E2E_CACHE[key] = _cpu_e2e(kind, distinct)
    return _CPU_E2E_CACHE[key]


def _cpu_e2e(kind, distinct):
    """The CPU side of SURVEY 8d (c): the same files from bytes to delivered PCM on the host cores -- oracle front-end +
    transform-stage oracle (FLAC since round 5: oracle/flac_frontend.c, no product code), one file per task on
    oracle/cpu_bench.c's native pthread pool, as many threads as the cgroup quota allows."""
    import ctypes as C
    import afgpu
    import oraclelib
    sys.path.insert(0, ROOT)
    from bench import host_cpu_info
    cpus, quota = host_cpu_info()
    threads = max(1, int(min(cpus, quota) if quota else cpus))
    codec = {"mp3": 10, "ogg": 11, "opus": 12, "flac": 13}[kind]
    keep, tasks = [], []
    for b in distinct:
        arr = np.frombuffer(b, np.uint8)
        keep.append(arr)
        t = oraclelib.BenchTask(codec, 0, 0, 0, 0, arr.ctypes.data, None, None, None, len(b))
        tasks.append(t)
    probe = tasks[:4]
    w1, _, s1 = oraclelib.bench_run(probe, 1, 1)              # single thread, a few files: sizes the run and gives the 1-thread rate
    single = s1 / w1
    per_pass = w1 / len(probe) * len(tasks) / threads            # estimated wall of one pass over all files
    reps = int(min(200, max(-(-4 * threads // len(tasks)), round(6.0 / max(per_pass, 1e-6)), 1)))
    wall, cpu_s, samples = oraclelib.bench_run(tasks, reps, threads)
    return {"value": samples / wall, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"{reps} passes over the {len(tasks)} distinct files of this batch, bytes -> delivered PCM (oracle front-end + transform oracle"
                      + ("; FLAC: oracle/flac_frontend.c, the reference's scalar reader with the prediction fused into the Rice loop" if kind == "flac" else "") + f"), {wall:.1f} s wall",
            "single_thread_value": single, "achieved_parallelism": cpu_s / wall}


E2E_CHECKED = 16          # distinct files of every end-to-end batch compared with the oracle's decode of the same bytes


def e2e_record(kind, name, distinct, files, threads, want_fn, tolerance=False, extra=None):
    blobs = batch_of(distinct, files)
    job, sec, windows = timed_batch(blobs, threads)
    usable = min(files, len(distinct))
    pick = sorted(set(int(round(k * (usable - 1) / max(1, E2E_CHECKED - 1))) for k in range(min(E2E_CHECKED, usable))))   # spread over the DISTINCT files
    items = [dict(o, pcm=o["pcm"].copy()) if k in pick else dict(o) for k, o in enumerate(job.items)]
    job.close()
    samples = sum(o["frames"] * o["channels"] for o in items)
    ok = all(o["status"] == 0 and o["frames"] > 0 for o in items)
    parts = [file_parity([items[k]], want_fn(distinct[k % len(distinct)]), tolerance) for k in pick]
    n = sum(r["samples"] for r in parts)
    sig = [r["rms_signal_per_file"][0] for r in parts]
    parity = {"files_checked": len(parts), "samples": n, "mismatches": sum(r["mismatches"] for r in parts),
              "rms_error": float(np.sqrt(sum(r["rms_error"] ** 2 * r["samples"] for r in parts) / max(n, 1))),
              "rms_error_worst_file": max(r["rms_error"] for r in parts),
              "rms_signal": float(np.sqrt(np.mean(np.square(sig)))), "rms_signal_per_file": [round(v, 5) for v in sig],
              "silent_files": sum(r.get("silent_files", 0) for r in parts)}
    for key in ("mode", "bitwise_mismatches", "int16_flip_rate"):
        if key in parts[0]:
            parity[key] = parts[0][key] if key == "mode" else (sum(r[key] for r in parts) if key == "bitwise_mismatches" else max(r[key] for r in parts))
    rec = {"workload": f"{files} x {name}: {len(distinct)} distinct generated files ({sum(map(len, distinct)) // len(distinct)} bytes on average), each its own buffer",
           "threads": threads, "all_ok": ok, "parity": parity, "seconds": sec, "seconds_per_call_windows": windows,
           "timing": f"median of {E2E_PASSES} windows of >= {E2E_WINDOW_S} s of back-to-back afg_batch_decode calls (the C call alone: no Python work between calls)",
           "samples_per_s_end_to_end": samples / sec, "samples_per_s_best_window": samples / min(windows), "compressed_MBps": sum(len(b) for b in blobs) / sec / 1e6,
           "cpu_baseline_e2e": cpu_e2e(kind, distinct)}
    rec["vs_cpu_baseline_e2e"] = rec["samples_per_s_end_to_end"] / rec["cpu_baseline_e2e"]["value"]
    rec.update(host_cpu_bound(samples, sec))
    if extra:
        rec.update(extra)
    return rec


def bench_flac_e2e(files, distinct, threads):
    """End to end through afg_batch_decode: file bytes in host memory -> host parse (threads) -> H2D -> restore kernel -> D2H ->
    interleaved floats in host memory.  Encoder-made files (tests/flac_bitstream.py: mid / side and left / side chosen per frame)."""
    import afgpu
    import oraclelib

    def want(data):                                      # the oracle from the bytes: its own front-end (oracle/flac_frontend.c)
        o = oraclelib.flac_decode_file(data)
        return (o["pcm"].astype(np.float64) * (1.0 / 2147483647.0)).astype(np.float32)          # stream.d:505-511
    # how many frames the host parser keeps as int16 residual rows (the headline C4 batch is all int16 rows)
    n16 = ntot = 0
    asg = {}
    for d in distinct[:64]:
        _, frames, _, res = afgpu.flac_parse(d)
        for fr in frames:                                # the batch path's rule (host/afg_flac_front.cpp): every residual and
            lo = int(fr["in_off"])                       # warm-up sample of the frame fits 16 bits, block of at least 8
            plane = res[lo:lo + int(fr["block_size"]) * int(fr["channels"])]
            n16 += int(int(fr["block_size"]) >= 8 and np.abs(plane.astype(np.int64)).max(initial=0) < 32768)
        ntot += len(frames)
        for a in frames["assignment"]:
            asg[int(a)] = asg.get(int(a), 0) + 1
    extra = {"res16_frame_fraction": n16 / max(ntot, 1), "channel_assignment_counts": asg,
             "res16_note": "share of frames whose residual rows the host parser ships as int16 (afg_flac_front.cpp: all residuals and warm-up samples fit 16 bits)"}
    return e2e_record("flac", "FLAC 16-bit stereo, frames of 4096, LPC order 8/12", distinct, files, threads, want, extra=extra)


def bench_mp3_e2e(files, distinct, threads):
    """MP3: file bytes -> host parse (sync, Huffman, scalefactors) -> H2D -> requantisation + transform kernels -> D2H."""
    import afgpu
    import oraclelib
    tol = afgpu.get_numeric_mode() == afgpu.NUMERIC_TOLERANCE
    return e2e_record("mp3", "MP3 128 kbit/s joint stereo", distinct, files, threads, lambda d: oraclelib.mp3_decode_file(d)["pcm"],
                      tolerance="rms" if tol else False)


def bench_vorbis_e2e(files, distinct, threads):
    """Ogg Vorbis: file bytes -> host parse (pages, code books, floor 1, residues) -> H2D -> coupling / floor + transform -> D2H."""
    import afgpu
    import oraclelib
    tol = afgpu.get_numeric_mode() == afgpu.NUMERIC_TOLERANCE
    return e2e_record("ogg", "Ogg Vorbis stereo 2048/256", distinct, files, threads,
                      lambda d: oraclelib.vorbis_file_pcm(oraclelib.vorbis_decode_file(d)), tolerance="rms" if tol else False)


def bench_opus_e2e(files, distinct, threads):
    """Ogg Opus (CELT-only): file bytes -> host parse (pages, framing, range decoder, CELT frame decoder) -> H2D -> transform
    kernels -> gain / int16 round trip -> D2H.  20 ms fullband stereo frames of 160 bytes (64 kbit/s), random payloads."""
    import afgpu
    import oraclelib
    tol = afgpu.get_numeric_mode() == afgpu.NUMERIC_TOLERANCE
    return e2e_record("opus", "Ogg Opus CELT-only stereo, 20 ms packets (R128 gain -78 dB: programme level)", distinct, files, threads,
                      lambda d: oraclelib.opus_file_pcm(oraclelib.opus_decode_file(d)), tolerance=tol)


def bench_device_inclusive(dev, chunks=8, files_per_chunk=16):
    """device-inclusive rate per codec -- transform-stage records already parsed, in page-locked host memory ->
    H2D + kernels + D2H, chunked so that the three overlap (upload + kernel on one stream, download on a second behind an
    event).  What the batch path's device stage costs once parsing is taken away; bound by the bytes that cross the bus in
    both directions together (74-89 GB/s on these boxes)."""
    import time
    from afgpu import corpus
    out = {}
    makers = {
        "mp3": lambda k: corpus.Mp3Part(0xA0D10 + k, np.full(files_per_chunk, corpus.C2_GRANULES), dev),
        "vorbis": lambda k: corpus.VorbisPart(0x0662 + k, np.full(files_per_chunk, corpus.C3_PACKETS), dev),
        "flac": lambda k: corpus.FlacPart(0xF1AC + k, np.full(4 * files_per_chunk, corpus.C4_FRAMES), dev),
    }
    inputs = {"mp3": lambda p: [p.coef, p.flags], "vorbis": lambda p: [p.spec], "flac": lambda p: [p.res]}
    up, down = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    for name, make in makers.items():
        parts = [make(k) for k in range(chunks)]
        torch.cuda.synchronize()
        h_in = [[t.cpu().pin_memory() for t in inputs[name](p)] for p in parts]
        h_out = [torch.empty(p.out_plane().shape, dtype=p.out_plane().dtype).pin_memory() for p in parts]
        evs = [torch.cuda.Event() for _ in parts]

        def one_pass():
            for k, p in enumerate(parts):
                with torch.cuda.stream(up):
                    for dst, src in zip(inputs[name](p), h_in[k]):
                        dst.copy_(src, non_blocking=True)
                    p.launch(up.cuda_stream)
                    evs[k].record(up)
                down.wait_event(evs[k])
                with torch.cuda.stream(down):
                    h_out[k].copy_(p.out_plane(), non_blocking=True)
            torch.cuda.synchronize()
        one_pass()
        walls = []
        for _ in range(5):
            t0 = time.perf_counter()
            one_pass()
            walls.append(time.perf_counter() - t0)
        walls.sort()
        sec = walls[len(walls) // 2]
        samples = sum(p.samples for p in parts)
        nbytes = sum(sum(t.numel() * t.element_size() for t in h) for h in h_in) + sum(t.numel() * t.element_size() for t in h_out)
        out[name] = {"workload": f"{chunks} chunks of {files_per_chunk * (4 if name == 'flac' else 1)} files at the headline shapes, records page-locked on the host",
                     "seconds": sec, "samples_per_s_device_inclusive": samples / sec, "bus_GBs_both_directions": nbytes / sec / 1e9,
                     "bytes_over_the_bus_per_sample": nbytes / samples}
        del parts, h_in, h_out
        torch.cuda.empty_cache()
    return out


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


def bench_mixed_e2e(files, gen, threads):
    """One afg_batch_decode call over a mix like BASELINE config C5 (40 % MP3, 25 % Ogg Vorbis, 25 % FLAC, 10 % Ogg Opus):
    distinct generated files, every one its own buffer."""
    import afgpu
    import oraclelib
    tol = afgpu.get_numeric_mode() == afgpu.NUMERIC_TOLERANCE
    pattern = ["mp3"] * 8 + ["ogg"] * 5 + ["flac"] * 5 + ["opus"] * 2
    count = {k: 0 for k in gen}
    blobs, kinds = [], []
    for i in range(files):
        k = pattern[i % len(pattern)]
        blobs.append(bytes(bytearray(gen[k][count[k] % len(gen[k])])))
        kinds.append((k, count[k] % len(gen[k])))
        count[k] += 1
    job, sec, windows = timed_batch(blobs, threads)
    picks = {}                                           # four DISTINCT files of every format: sixteen per batch
    for i, (k, j) in enumerate(kinds):
        lst = picks.setdefault(k, [])
        if len(lst) < 4 and all(j != jj for _, jj in lst):
            lst.append((i, j))
    chosen = {i for lst in picks.values() for i, _ in lst}
    items = [dict(o, pcm=o["pcm"].copy()) if i in chosen else dict(o) for i, o in enumerate(job.items)]
    job.close()
    samples = sum(o["frames"] * o["channels"] for o in items)
    ok = all(o["status"] == 0 and o["frames"] > 0 for o in items)

    def flac_want(d):
        o = oraclelib.flac_decode_file(d)
        return (o["pcm"].astype(np.float64) * (1.0 / 2147483647.0)).astype(np.float32)
    wants = {"mp3": lambda d: oraclelib.mp3_decode_file(d)["pcm"], "ogg": lambda d: oraclelib.vorbis_file_pcm(oraclelib.vorbis_decode_file(d)),
             "flac": flac_want, "opus": lambda d: oraclelib.opus_file_pcm(oraclelib.opus_decode_file(d))}
    parity = {}
    for k, lst in picks.items():
        mode = tol if k == "opus" else ("rms" if (tol and k in ("ogg", "mp3")) else False)
        parts = [file_parity([items[i]], wants[k](gen[k][j]), mode) for i, j in lst]
        n = sum(r["samples"] for r in parts)
        parity[k] = {"files_checked": len(parts), "samples": n, "mismatches": sum(r["mismatches"] for r in parts),
                     "rms_error": float(np.sqrt(sum(r["rms_error"] ** 2 * r["samples"] for r in parts) / max(n, 1))),
                     "rms_signal_per_file": [round(r["rms_signal_per_file"][0], 5) for r in parts],
                     "silent_files": sum(r.get("silent_files", 0) for r in parts)}
        if "mode" in parts[0]:
            parity[k]["mode"] = parts[0]["mode"]
    per = {}
    for (k, _), o in zip(kinds, items):
        per[k] = per.get(k, 0) + o["frames"] * o["channels"]
    # CPU side: the four codecs' cpu_baseline_e2e rates combined in this batch's sample proportions (harmonic mean)
    cpu = {k: cpu_e2e(k, gen[k]) for k in per}
    cpu_rate = samples / sum(per[k] / cpu[k]["value"] for k in per)
    return {**host_cpu_bound(samples, sec),
            "workload": f"{files} mixed files in one batch (40 % MP3, 25 % Ogg Vorbis, 25 % FLAC, 10 % Ogg Opus), {sum(len(v) for v in gen.values())} distinct, each its own buffer",
            "threads": threads, "all_ok": ok, "parity": parity, "seconds": sec, "seconds_per_call_windows": windows,
            "timing": f"median of {E2E_PASSES} windows of >= {E2E_WINDOW_S} s of back-to-back afg_batch_decode calls",
            "samples": samples, "samples_by_format": per, "samples_per_s_end_to_end": samples / sec,
            "compressed_MBps": sum(len(b) for b in blobs) / sec / 1e6,
            "cpu_baseline_e2e": {"value": cpu_rate, "unit": "samples/s", "cores": next(iter(cpu.values()))["cores"], "kind": "port",
                                 "sample": "the per-codec cpu_baseline_e2e rates combined in this batch's sample proportions"},
            "vs_cpu_baseline_e2e": (samples / sec) / cpu_rate}


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
    ap.add_argument("--e2e-distinct", type=int, default=E2E_DISTINCT)
    args = ap.parse_args()
    # the generated files of the end-to-end legs come from a process pool: before anything initialises the GPU
    gen = generate_files({"mp3": 60, "ogg": 128, "flac": 8, "opus": 250}, args.e2e_distinct) if args.codec == "others" else None
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
    for name, kind, size, fn in (("mp3_e2e", "mp3", 60, bench_mp3_e2e), ("vorbis_e2e", "ogg", 128, bench_vorbis_e2e),
                                 ("opus_e2e", "opus", 250, bench_opus_e2e), ("flac_e2e", "flac", 8, bench_flac_e2e)):
        if args.codec == name:
            res[name] = fn(args.e2e_files, generate_files({kind: size}, args.e2e_distinct)[kind], args.e2e_threads)
    if args.codec == "mixed_e2e":
        res["mixed_e2e"] = bench_mixed_e2e(args.e2e_files * 2, generate_files({"mp3": 60, "ogg": 64, "flac": 16, "opus": 125}, args.e2e_distinct),
                                           args.e2e_threads)
    if args.codec == "qoa_enc":
        res["qoa_enc"] = bench_qoa_encode(dev, 8192, 4.0, args.steps, args.warmup)
    if args.codec == "device_inclusive":
        res["device_inclusive"] = bench_device_inclusive(dev)
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
        res["device_inclusive"] = bench_device_inclusive(dev)
        torch.cuda.empty_cache()
        res["mp3_e2e"] = bench_mp3_e2e(args.e2e_files, gen["mp3"], args.e2e_threads)
        res["vorbis_e2e"] = bench_vorbis_e2e(args.e2e_files, gen["ogg"], args.e2e_threads)
        res["flac_e2e"] = bench_flac_e2e(args.e2e_files, gen["flac"], args.e2e_threads)
        res["opus_e2e"] = bench_opus_e2e(args.e2e_files, gen["opus"], args.e2e_threads)
        res["mixed_e2e"] = bench_mixed_e2e(args.e2e_files * 2, gen, args.e2e_threads)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
