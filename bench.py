#!/usr/bin/env python3
"""bench.py -- decoded samples/s of the batched transform stage on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (for N > 1 launched by
torch.distributed.run, one rank per GPU).  A step is one pass of the hot path over one
device-resident batch: BASELINE.json configs[1], 1024 MP3 CBR-128k stereo files of 60 s
(2 297 frames x 2 granules x 2 channels x 576 lines each), i.e. one launch of the MP3
transform kernel.  Files are independent, so N GPUs each own a 1024-file shard (weak
scaling, no collective on the data path); `value` is the samples all ranks decoded
divided by the slowest rank's time.

Rank 0 prints ONE JSON line with the `roofline` of the dominant kernel (measured with
events on the launch stream inside the timed region) and the `cpu_baseline`: the CPU
oracle (a scalar C port of the reference algorithms; the D reference cannot be built
here) timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "audio-formats_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
MP3_BYTES_PER_GRCH = 2304 + 4 + 2304   # algorithmic bytes: f32 spectrum + flag word in, f32 PCM out
FRAMES_PER_FILE = 2297           # 60 s of 128 kbps MPEG-1 Layer III at 44.1 kHz
GRANULES_PER_FILE = 2 * FRAMES_PER_FILE


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--files", type=int, default=1024, help="files per GPU (BASELINE config: 1024)")
    ap.add_argument("--seg", type=int, default=0, help="granules per wavefront segment (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--no-extra", action="store_true", help="skip the Vorbis (C3) / FLAC (C4) kernel timings")
    return ap.parse_args()


def cpu_baseline(coef_files, flag_files, seconds):
    """Oracle (scalar C port of minimp3.d's transform stage) on the host cores: one file per task."""
    import threading
    import oraclelib
    oraclelib.lib()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    nfiles = len(coef_files)
    granules = np.array([GRANULES_PER_FILE], np.uint32)
    channels = np.array([2], np.uint8)
    tls = threading.local()

    def one(i):
        if not hasattr(tls, "pcm"):
            tls.pcm = np.empty(coef_files[0].size, np.float32)      # per-thread output, no allocation while timing
        oraclelib.mp3_transform_into(granules, channels, coef_files[i % nfiles], flag_files[i % nfiles], tls.pcm)
        return tls.pcm.size

    one(0)                                              # page in
    t0 = time.perf_counter()
    n1 = one(0)
    t1 = time.perf_counter() - t0                       # one file, one thread
    per_pass = max(nfiles, cores)
    with ThreadPoolExecutor(max_workers=cores) as pool:
        t0 = time.perf_counter()
        sum(pool.map(one, range(per_pass)))             # untimed-for-the-result pass: sizes the run
        pass_s = time.perf_counter() - t0
        passes = max(1, min(1000, int(seconds / max(pass_s, 1e-3))))
        t0 = time.perf_counter()
        done = sum(pool.map(one, range(per_pass * passes)))
        dt = time.perf_counter() - t0
    return {
        "value": done / dt, "unit": "samples/s", "cores": cores, "kind": "port",
        "sample": f"{per_pass * passes} file-decodes ({nfiles} distinct 60 s stereo files of the same batch, "
                  f"{passes} passes) through oracle/mp3_transform.c, one file per thread task, {dt:.1f} s wall",
        "single_thread_value": n1 / t1,
    }


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import afgpu
    from afgpu import synthetic
    afgpu.lib()

    # ---- device-resident synthetic batch (this rank's shard of files) -----------------
    n_files = args.files
    granules = np.full(n_files, GRANULES_PER_FILE, np.uint32)
    channels = np.full(n_files, 2, np.uint8)
    seed = 0xA0D10 + 7919 * rank
    coef, flags = synthetic.mp3_batch_device(seed, n_files, GRANULES_PER_FILE, dev)
    plan = afgpu.Mp3Plan(granules, channels, args.seg)
    assert plan.blocks * 576 == coef.numel()
    pcm = torch.empty_like(coef)
    stream = torch.cuda.current_stream()
    samples_per_step = plan.blocks * 576
    coef_numel = coef.numel()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        plan.transform(coef, flags, pcm, None, stream)
    barrier()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        starts[i].record(stream)                       # same stream the kernel is launched on
        plan.transform(coef, flags, pcm, None, stream)
        ends[i].record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = [s.elapsed_time(e) for s, e in zip(starts, ends)]

    # device-to-device copy of the same byte volume (spectrum plane -> PCM plane) with the library's streaming
    # copy kernel: the copy rate this box sustains for a read-N/write-N stream, reported next to the 8 TB/s spec
    # the roofline is priced against (hipMemcpy / torch copy_ reach only ~4.6-4.9 TB/s on the same buffers)
    copy_ms = []
    if rank == 0:
        scratch = torch.empty_like(coef)
        for i in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            afgpu.copy_probe(scratch, coef, coef.numel() * 4)
            e1.record(stream)
            torch.cuda.synchronize()
            if i:
                copy_ms.append(e0.elapsed_time(e1))
        del scratch

    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- parity of this very output against the oracle on the first files --------------
    parity = None
    cpu = None
    if rank == 0:
        import oraclelib
        per_file = GRANULES_PER_FILE * 2 * 576
        ncheck = min(2, n_files)
        c_h = coef[:ncheck * per_file].cpu().numpy()
        f_h = flags[:ncheck * GRANULES_PER_FILE * 2].cpu().numpy().view(np.uint32)
        got = pcm[:ncheck * per_file].cpu().numpy()
        want = oraclelib.mp3_transform(granules[:ncheck], channels[:ncheck], c_h, f_h)
        diff = got.astype(np.float64) - want.astype(np.float64)
        parity = {"files_checked": ncheck, "samples": int(got.size),
                  "bitwise_mismatches": int((got.view(np.uint32) != want.view(np.uint32)).sum()),
                  "rms_error": float(np.sqrt(np.mean(diff ** 2))), "max_abs_error": float(np.abs(diff).max())}
        if not args.no_cpu_baseline and world == 1:
            nsample = min(n_files, max(8, min(64, os.cpu_count() or 1)))
            c_s = coef[:nsample * per_file].cpu().numpy().reshape(nsample, -1)
            f_s = flags[:nsample * GRANULES_PER_FILE * 2].cpu().numpy().view(np.uint32).reshape(nsample, -1)
            cpu = cpu_baseline([c_s[i] for i in range(nsample)], [f_s[i] for i in range(nsample)],
                               args.cpu_seconds)

    if rank != 0:
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        return

    total_samples = samples_per_step * args.steps * world
    value = total_samples / elapsed
    avg_kernel_s = (sum(kernel_ms) / len(kernel_ms)) * 1e-3
    alg_bytes = plan.blocks * MP3_BYTES_PER_GRCH
    achieved = alg_bytes / avg_kernel_s / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_mp3.json")
    if os.path.exists(tpath):
        try:
            with open(tpath) as fh:
                tj = json.load(fh)
            if tj.get("files") == n_files and tj.get("seg", 0) == args.seg:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    # ---- the other two codecs of the metric (BASELINE configs[2], configs[3]), kernel-level, N = 1 only ----
    other = None
    if world == 1 and not args.no_extra:
        del coef, pcm, flags
        torch.cuda.empty_cache()
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_codecs
        other = {}
        try:
            other["vorbis_c3"] = bench_codecs.bench_vorbis(dev, 1024, 2584, 3, 1, 0)
            torch.cuda.empty_cache()
            other["flac_c4"] = bench_codecs.bench_flac(dev, 4096, 323, 3, 1, False)
        except Exception as e:          # the headline line must still print
            other["error"] = repr(e)

    line = {
        "metric": "decoded samples/sec (batched MP3+OGG+FLAC) at 1/2/4/8 MI355X vs CPU ref",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{n_files}-file batched MP3 CBR-128k stereo per GPU "
                               "(L3 antialias+imdct36/12 + DCT-II + polyphase synth kernel), 60 s files, "
                               "device-resident dequantised spectra -> interleaved f32 PCM",
                   "files_per_gpu": n_files, "granule_channels_per_gpu": int(plan.blocks),
                   "samples_per_step_per_gpu": int(samples_per_step), "segments": plan.segments,
                   "parallelism": f"file-sharded x{world}, no collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "mp3_transform_kernel", "avg_kernel_ms": avg_kernel_s * 1e3,
                     "measured_copy_GBs": (2 * 4 * coef_numel / (min(copy_ms) * 1e-3) / 1e9) if copy_ms else None,
                     "algorithmic_bytes_per_launch": int(alg_bytes)},
        "cpu_baseline": cpu,
        "parity": parity,
        "other_workloads": other,
    }
    print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
