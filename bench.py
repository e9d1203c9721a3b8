#!/usr/bin/env python3
"""bench.py -- decoded samples/s of the batched transform stage on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`.  With N > 1 and no WORLD_SIZE in the
environment this process starts N rank processes itself (one per GPU, before anything here touches a
GPU) and relays rank 0's line; under `torch.distributed.run` it is one of the N ranks.  Ranks only
meet in a `gloo` barrier and a MAX-reduce of the elapsed time: files are independent, so there is
no collective on the data path and no RCCL.

A *step* is one pass of the hot path over the device-resident batches of the configuration:

  --config c234 (default)  BASELINE.json's metric "batched MP3+OGG+FLAC": configs[1] + [2] + [3] resident
                           together -- 1024 x 60 s MP3 (C2), 1024 x 2584-packet Ogg Vorbis (C3),
                           4096 x 323-frame FLAC (C4) -- one launch of each codec's kernel(s) per step.
                           Every rank owns such a set (weak scaling).
  --config c2 | c3 | c4    one of them alone.
  --config c5              configs[4]: the 65 536-file mixed MP3 / Vorbis / FLAC / Opus-CELT corpus sharded by file
                           over the ranks (strong scaling: the same corpus at every N), walked in waves of
                           <= 24576 files that fit one GPU; a step is one pass over the rank's whole shard.

Rank 0 prints ONE JSON line with `roofline` (per-kernel event times measured on the launch stream
inside the timed region; the dominant kernel at the top level) and `cpu_baseline` (the CPU oracle, a
scalar C port of the reference algorithms -- the D reference cannot be built here -- timed by a native
thread pool on this box's host cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "audio-formats_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
METRIC = "decoded samples/sec (batched MP3+OGG+FLAC) at 1/2/4/8 MI355X vs CPU ref"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c234", choices=["c234", "c2", "c3", "c4", "c5"])
    ap.add_argument("--files", type=int, default=1024, help="C2/C3 files per GPU (C4 has 4x as many); BASELINE: 1024")
    ap.add_argument("--c5-files", type=int, default=65536, help="files of the mixed corpus (BASELINE: 65536)")
    ap.add_argument("--seg", type=int, default=0, help="MP3 granules per wavefront segment (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-full-fetch", action="store_true", help="skip the MP3 leg that fetches all 32 subbands")
    ap.add_argument("--c5-wave-files", type=int, default=0, help="files per resident wave of --config c5 (0 = the corpus default)")
    ap.add_argument("--only", default="", help="development: restrict --config c5 to these codecs (comma list)")
    ap.add_argument("--no-others", action="store_true",
                    help="skip `other_workloads` (the side-by-side step; the C5 corpus, dense CELT, QOA and the end-to-end batches, each in "
                         "a child process after the headline measurement; only at N = 1 with the default config)")
    ap.add_argument("--measure-traffic", action="store_true", default=None,
                    help="N = 1, default config (where it is the default when rocprofv3 is on the PATH): before the run, measure `roofline.traffic` of "
                         "the three headline kernels in child processes (separate `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` passes over bench.py "
                         "--config c2 / c3 / c4, FLAC's counters calibrated on its access pattern: tools/pmc_collect.sh) instead of reading the "
                         "committed passes under profiles/; adds about half a minute")
    ap.add_argument("--no-measure-traffic", dest="measure_traffic", action="store_false",
                    help="read `roofline.traffic` from the committed passes under profiles/ (labelled `traffic_from`)")
    ap.add_argument("--full-line", action="store_true",
                    help="print the full record as the (only) stdout line instead of the compact line (what this script's own child "
                         "runs and the tools that post-process a run read)")
    ap.add_argument("--full-record", default="", help="where the full record goes (default: gpurun_out/bench_full.json under the repo)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="testing only: ranks beyond the visible devices share them (rank %% devices); the line says so")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------
# --measure-traffic: HBM bytes per launch from the PMC counters, measured in child processes before this run
# ----------------------------------------------------------------------------------------------------------------
def measure_traffic():
    """{codec: hbm bytes per launch} from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters only -- never combined
    with tracing) over `bench.py --config cN`, each in its own child process; this process has not touched a GPU yet.  FLAC's
    counters are calibrated first on the kernel's own access pattern with known byte counts (MI355X_MICROARCH.md asks for that
    before trusting an access width it does not list).  Returns {} with a message on stderr when rocprofv3 or hipcc is missing."""
    import shutil
    if not shutil.which("rocprofv3"):
        sys.stderr.write("bench.py --measure-traffic: rocprofv3 not found; falling back to the committed passes\n")
        return {}
    env = dict(os.environ, GRAFT_REPO_ROOT=ROOT, AFG_PMC_SETS="FETCH_SIZE;WRITE_SIZE", AFG_BENCH_NESTED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = {}
    common = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-full-fetch", "--no-others"]
    jobs = [("mp3", "mp3_tolerance_kernel", "c2", {}), ("vorbis", "vorbis_walk_kernel", "c3", {})]
    try:
        subprocess.run(["bash", os.path.join(ROOT, "tools", "pmc_calib_flac.sh")], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
        with open(os.path.join(ROOT, "gpurun_out", "calib_flac", "calib.json")) as fh:
            c = json.load(fh)
        f = sum(c["FETCH_SIZE_kb_per_launch"]) / len(c["FETCH_SIZE_kb_per_launch"]) * 1024
        w = sum(c["WRITE_SIZE_kb_per_launch"]) / len(c["WRITE_SIZE_kb_per_launch"]) * 1024
        jobs.append(("flac", "flac_restore1_kernel", "c4", {"AFG_PMC_FETCH_FACTOR": f"{c['known_read_bytes'] / f:.4f}",
                                                           "AFG_PMC_WRITE_FACTOR": f"{c['known_write_bytes'] / w:.4f}", "AFG_PMC_DISPATCHES_PER_LAUNCH": "2"}))
    except Exception as e:
        sys.stderr.write(f"bench.py --measure-traffic: FLAC calibration failed ({e}); FLAC keeps the committed pass\n")
    for codec, needle, cfg, extra in jobs:
        tag = f"live_pmc_{codec}"
        try:
            subprocess.run(["bash", os.path.join(ROOT, "tools", "pmc_collect.sh"), tag, needle, "bench.py", "--config", cfg] + common,
                           env=dict(env, **extra), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=150)
            with open(os.path.join(ROOT, "gpurun_out", tag, tag + ".json")) as fh:
                out[codec] = float(json.load(fh)["derived"]["hbm_bytes_per_launch"])
        except Exception as e:
            sys.stderr.write(f"bench.py --measure-traffic: {codec}: {e}\n")
    return out

# ----------------------------------------------------------------------------------------------------------------
# launcher: N rank processes, started before this process has touched a GPU (it never does)
# ----------------------------------------------------------------------------------------------------------------
def launch_ranks(args):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AFG_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    bad = [c for c in codes if c]
    return bad[0] if bad else 0


# ----------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle under a native thread pool (oracle/cpu_bench.c)
# ----------------------------------------------------------------------------------------------------------------
def host_cpu_info():
    """Logical CPUs this process may run on, and the cgroup CPU quota (None = unlimited / unknown)."""
    cpus = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                         # cgroup v2
            q, per = fh.read().split()
            if q != "max":
                quota = float(q) / float(per)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:        # cgroup v1
                q = float(fh.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                per = float(fh.read())
            if q > 0:
                quota = q / per
        except Exception:
            quota = None
    return cpus, quota


def cpu_tasks_from_parts(parts, files_per_part):
    """Oracle tasks for the first files of each device-resident part (inputs copied to host once)."""
    import numpy as np
    import oraclelib
    tasks, keep, desc = [], [], []
    for p in parts:
        nfile = files_per_part.get(p.name, 0)
        if p.name == "mp3":
            off = 0
            for f in range(nfile):
                ng = int(p.granules[f])
                nb = ng * 2
                coef = p.coef[off * 576:(off + nb) * 576].cpu().numpy()
                flags = p.flags[off:off + nb].cpu().numpy().view(np.uint32).copy()
                keep += [coef, flags]
                tasks.append(oraclelib.bench_task(0, ng, 2, 0, 0, coef, flags, None, None, nb * 576))
                off += nb
        elif p.name == "vorbis":
            so, oo = p.plan.offsets()
            pk = 0
            for f in range(nfile):
                npk = int(p.plan.packets[f])
                s0 = int(so[pk])
                s1 = int(so[pk + npk]) if pk + npk < p.plan.total_packets else p.plan.spec_floats
                o0 = int(oo[pk])
                o1 = int(oo[pk + npk]) if pk + npk < p.plan.total_packets else p.plan.out_floats
                spec = p.spec[s0:s1].cpu().numpy()
                pf = p.plan.pflags[pk:pk + npk].copy()
                soff = (so[pk:pk + npk] - np.uint64(s0)).astype(np.uint64)
                ooff = (oo[pk:pk + npk] - np.uint64(o0)).astype(np.uint64)
                keep += [spec, pf, soff, ooff]
                tasks.append(oraclelib.bench_task(1, npk, 2, int(p.plan.bs0[f]), int(p.plan.bs1[f]), spec, pf, soff, ooff, o1 - o0))
                pk += npk
        elif p.name == "flac":
            fr0 = 0
            for f in range(nfile):
                nfr = int(p.frames_per_file[f])
                frames = p.frames_np[fr0:fr0 + nfr].copy()
                w0, o0 = int(frames["in_off"][0]), int(frames["out_off"][0])
                words = nfr * 2 * p.block_size
                frames["in_off"] -= np.uint64(w0)
                frames["out_off"] -= np.uint64(o0)
                frames["sf_index"] -= np.uint32(2 * fr0)
                sub = p.sub_np[2 * fr0:2 * (fr0 + nfr)].copy()
                if p.res16:
                    # int16 residual rows (in_off counts int16 elements of the int32-typed plane): the CPU leg gets them
                    # widened beforehand, as the int32 rows the reference's decoder holds -- no widening pass in its timing
                    assert p.block_size % 8 == 0
                    res = p.res[w0 // 2:(w0 + words) // 2].cpu().numpy().view(np.int16).astype(np.int32)
                    frames["res16"] = 0
                else:
                    res = p.res[w0:w0 + words].cpu().numpy()
                keep += [frames, sub, res]
                tasks.append(oraclelib.bench_task(2, nfr, 2, 0, 0, frames, sub, res, None, words))
                fr0 += nfr
        elif p.name == "celt":
            r0 = 0
            for f in range(nfile):
                nfr = int(p.frames_per_file[f])
                nrec = 2 * nfr
                recs = p.recs_np[r0:r0 + nrec].copy()
                c0, o0 = int(recs["coef_off"].min()), int(recs["out_off"].min())
                recs["coef_off"] -= np.uint64(c0)
                recs["out_off"] -= np.uint64(o0)
                rb = np.array([0, nfr, nrec], np.uint64)
                coef = p.coef[c0:c0 + nrec * 960].cpu().numpy()
                keep += [recs, rb, coef]
                tasks.append(oraclelib.bench_task(3, 2, 2, 0, 0, rb, recs, coef, None, nrec * 960))
                r0 += nrec
        desc.append(f"{nfile} {p.name}")
    return tasks, keep, desc


def cpu_baseline(parts, seconds):
    """The oracle on the host cores, one file per task, files in the workload's own proportions."""
    import oraclelib
    cpus, quota = host_cpu_info()
    threads = max(1, int(min(cpus, quota) if quota else cpus))
    # files per pass in the proportion the workload holds them (C2:C3:C4 = 1:1:4), scaled to give every thread work
    counts = {p.name: 0 for p in parts}
    nfiles = {"mp3": lambda p: len(p.granules), "vorbis": lambda p: len(p.plan.packets),
              "flac": lambda p: len(p.frames_per_file), "celt": lambda p: len(p.frames_per_file)}
    least = min(nfiles[p.name](p) for p in parts)
    for p in parts:
        ratio = max(1, round(nfiles[p.name](p) / least))
        counts[p.name] = min(nfiles[p.name](p), ratio * max(2, min(16, threads // 8)))
    tasks, keep, desc = cpu_tasks_from_parts(parts, counts)
    t1_wall, t1_cpu, t1_samples = oraclelib.bench_run(tasks, 1, 1)            # single thread, one pass
    single = t1_samples / t1_wall
    est_pass = t1_wall / threads
    # size the run: whole passes, each pass must offer >= 4 tasks per thread so the tail does not dominate
    reps_min = max(1, -(-4 * threads // len(tasks)))
    reps = max(reps_min, int(seconds / max(est_pass, 1e-6)))
    reps = min(reps, max(reps_min, int(3 * seconds / max(est_pass, 1e-6))))
    wall, cpu_s, samples = oraclelib.bench_run(tasks, reps, threads)
    return {
        "value": samples / wall, "unit": "samples/s", "cores": threads, "kind": "port",
        "sample": f"{reps} passes over {' + '.join(desc)} files of this workload (one file per task, native pthread pool, "
                  f"oracle/*.c scalar -O2), {wall:.1f} s wall",
        "single_thread_value": single,
        "logical_cpus": cpus, "cgroup_cpu_quota": quota,
        "achieved_parallelism": cpu_s / wall, "parallel_efficiency_vs_single_thread": (samples / wall) / (single * threads),
    }


# ----------------------------------------------------------------------------------------------------------------
# other workloads: everything SURVEY 8d asks for besides the headline step, under the same driver clock
# ----------------------------------------------------------------------------------------------------------------
def other_workloads(args):
    """Run after the headline measurement has released the device, each in a child process (a fresh HIP context; the C5
    waves need the memory the headline batches held):
      c5          BASELINE configs[4] at full size on this one GPU: 65 536 mixed files in 3 resident waves (bench.py --config c5)
      flac_int32_rows  C4 with int32 residual rows (8 B per sample moved) beside the headline's int16 rows
      celt_dense  8192 x Opus/CELT stereo, 200 frames of 960: the CELT kernel on a device-filling batch
      qoa         4096 x QOA stereo 4 s
      vorbis_shapes  C3-sized batches of the other Vorbis stream shapes (mono, blocksize_1 1024 / 4096, 3 and 6 channels)
      flac_shapes    C4-sized batches of other FLAC stream shapes (24-bit, LPC order 32, mono, 6 channels, blocks of 1152 / 576)
      device_inclusive  SURVEY 8d (b): parsed records in page-locked memory -> H2D + kernels + D2H, overlapped, per headline codec
      *_e2e       SURVEY 8d (c): file bytes in host memory -> afg_batch_decode (host parse, H2D, kernels, D2H) -> floats in
                  host memory; PCIe-inclusive, never `value`.  256 DISTINCT generated files per codec (every blob its own
                  buffer), median of 5 windows of >= 1 s; beside each rate `cpu_baseline_e2e`: the same files from bytes to
                  PCM through the oracle front-ends on the host cores (native thread pool, the cgroup quota's threads)
    Every entry carries its own parity block against the oracle.  At N > 1 (default config) the same key holds
    `c5` only: BASELINE configs[4] strong-scaled over the ranks that just ran the headline step."""
    out = {}
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    def child(cmd, timeout):
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable] + cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
        except subprocess.TimeoutExpired:
            return None, f"timed out after {timeout} s", time.perf_counter() - t0
        lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
        if not lines:
            return None, f"exit code {r.returncode}: {r.stderr.decode()[-400:]}", time.perf_counter() - t0
        return json.loads(lines[-1]), (None if r.returncode == 0 else f"exit code {r.returncode}"), time.perf_counter() - t0

    d, err, wall = child([os.path.abspath(__file__), "--config", "c5", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--full-line",
                          "--c5-files", str(args.c5_files)], 420)
    if d is None:
        out["c5"] = {"error": err}
    else:
        out["c5"] = {"workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
                     "steps": d["steps"], "samples_per_step": d["config"]["samples_per_step"], "waves": d["config"].get("waves_per_gpu"),
                     "kernels": [{k: v for k, v in kk.items() if k in ("codec", "kernel", "avg_kernel_ms", "achieved", "frac", "samples_per_launch",
                                                                       "algorithmic_bytes_per_launch")} for kk in d["roofline"]["kernels"]],
                     "overlapped_on_a_second_stream": d["roofline"].get("overlapped_on_a_second_stream"),
                     "celt_alone": d["roofline"].get("celt_alone"),
                     "parity": d["parity"], "wall_s": wall, "error": err}
    # C4 with the residual rows left as int32 (what material with 17-bit warm-up samples falls back to, frame by frame)
    env32 = dict(env, AFG_FLAC_RES32="1")
    t0 = time.perf_counter()
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", "c4", "--steps", "5", "--warmup", "1", "--no-cpu-baseline",
                            "--no-others", "--full-line", "--files", str(args.files)], env=env32, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
        d = json.loads(lines[-1]) if lines else None
        err = None if (d is not None and r.returncode == 0) else f"exit code {r.returncode}: {r.stderr.decode()[-300:]}"
    except subprocess.TimeoutExpired:
        d, err = None, "timed out after 300 s"
    if d is None:
        out["flac_int32_rows"] = {"error": err}
    else:
        k = d["roofline"]["kernels"][0]
        out["flac_int32_rows"] = {"workload": d["config"]["workload"], "avg_kernel_ms": k["avg_kernel_ms"], "achieved": k["achieved"], "frac": k["frac"],
                                  "algorithmic_bytes_per_launch": k["algorithmic_bytes_per_launch"], "samples_per_s": k["samples_per_s"],
                                  "parity": d["parity"], "wall_s": time.perf_counter() - t0, "error": err}
    d, err, wall = child([os.path.join(ROOT, "tools", "bench_codecs.py"), "--codec", "others", "--steps", "5"], 900)
    if d is None:
        out["codecs"] = {"error": err}
    else:
        for k, v in d.items():
            out[k] = v
        out["codecs_wall_s"] = wall
    # the other stream shapes of the Vorbis walk (mono, 1024 / 4096-sample long blocks, more than two channels), C3-sized
    d, err, wall = child([os.path.join(ROOT, "tools", "vorbis_shapes.py"), "--steps", "5", "--files", str(args.files)], 300)
    out["vorbis_shapes"] = {"error": err} if d is None else dict(d["vorbis_shapes"], wall_s=wall, error=d["vorbis_shapes"]["error"] or err)
    # ... and of the FLAC restore (24-bit material with wide sums, LPC order 32, mono, six channels, short blocks), C4-sized
    d, err, wall = child([os.path.join(ROOT, "tools", "flac_shapes.py"), "--steps", "5"], 300)
    out["flac_shapes"] = {"error": err} if d is None else dict(d["flac_shapes"], wall_s=wall, error=d["flac_shapes"]["error"] or err)
    return out


def other_parity_failures(others):
    bad = []
    for name, rec in others.items():
        if not isinstance(rec, dict):
            continue
        if rec.get("error"):
            bad.append(f"{name}: {rec['error']}")
        par = rec.get("parity")
        if isinstance(par, dict):
            blocks = par.values() if all(isinstance(v, dict) for v in par.values()) else [par]
            if any(b.get("mismatches") for b in blocks):
                bad.append(f"{name}: parity")
        for key in ("mismatches", "int32_mismatches"):
            if rec.get(key):
                bad.append(f"{name}: {key}")
        if rec.get("numeric_mode") == "exact" and rec.get("bitwise_mismatches"):
            bad.append(f"{name}: bitwise_mismatches")
        if rec.get("numeric_mode") == "tolerance" and not rec.get("rms_vs_oracle", 0.0) <= 1e-5:
            bad.append(f"{name}: rms_vs_oracle")
        if rec.get("all_ok") is False:
            bad.append(f"{name}: not all files decoded")
    return bad



# ----------------------------------------------------------------------------------------------------------------
# the printed line: compact (what the driver parses); the full record goes to a file
# ----------------------------------------------------------------------------------------------------------------
COMPACT_LIMIT = 6000            # characters; the driver keeps an 8 KB tail of stdout and parses its last JSON line


def _sig(x, digits=5):
    """Numbers to `digits` significant figures (ints and everything else unchanged): the line is a summary."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float(f"{x:.{digits}g}")


def _pick(d, keys):
    return {k: _sig(d[k]) for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(full, full_record_path=None):
    """The one JSON line `bench.py` prints: the contract's keys, `roofline`, `cpu_baseline`, parity per codec and ONE
    number (or a short tuple of numbers) per `other_workloads` leg.  Everything else -- notes, per-window timings, per-file
    lists, workload prose -- is in the full record (`full_record_path`).  Never longer than COMPACT_LIMIT characters:
    optional blocks are dropped from the end until it fits."""
    r = full.get("roofline") or {}
    out = {k: _sig(full.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                          "scaling", "vs_baseline", "dtype", "data")}
    cfg = full.get("config") or {}
    out["config"] = {k: cfg[k] for k in ("workload", "name", "samples_per_step", "files_per_gpu", "files", "waves_per_gpu", "parallelism") if k in cfg}
    if len(out["config"].get("workload", "")) > 330:
        out["config"]["workload"] = out["config"]["workload"][:327] + "..."
    roof = _pick(r, ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_from", "frac_by_traffic", "kernel", "codec", "avg_kernel_ms",
                     "algorithmic_bytes_per_launch", "measured_copy_GBs"))
    if "kernel" in roof:
        roof["kernel"] = str(roof["kernel"]).split(" ")[0]
    roof.setdefault("traffic", None)
    roof["kernels"] = [dict(_pick(k, ("codec", "avg_kernel_ms", "frac", "traffic", "algorithmic_bytes_per_launch", "samples_per_launch")),
                            kernel=str(k.get("kernel", "")).split(" ")[0]) for k in r.get("kernels", [])]
    if r.get("whole_step"):
        roof["whole_step"] = _pick(r["whole_step"], ("algorithmic_bytes", "kernel_ms", "frac"))
    for leg in ("mp3_full_fetch", "vorbis_full_fetch", "celt_alone"):
        if r.get(leg):
            roof[leg] = _pick(r[leg], ("avg_kernel_ms", "frac"))
    out["roofline"] = roof
    cpu = full.get("cpu_baseline")
    out["cpu_baseline"] = None if cpu is None else dict(_pick(cpu, ("value", "unit", "cores", "kind", "single_thread_value")),
                                                        sample=str(cpu.get("sample", ""))[:160])
    out["parity"] = {c: _pick(p, ("mismatches", "rms_error", "rms_signal", "samples")) for c, p in (full.get("parity") or {}).items()}
    if full.get("oversubscribed"):
        out["oversubscribed"] = full["oversubscribed"]
    ow = full.get("other_workloads") or {}
    others = {}
    for name, rec in ow.items():
        if not isinstance(rec, dict):
            continue
        if rec.get("error") and len([k for k in rec if k != "error"]) == 0:
            others[name] = {"error": str(rec["error"])[:120]}
            continue
        if name == "c5":
            e = _pick(rec, ("value", "ms_per_step", "samples_per_step", "n_gpus", "efficiency_vs_n1"))
            e["kernel_ms"] = {k.get("codec"): _sig(k.get("avg_kernel_ms"), 4) for k in rec.get("kernels", [])}
            if isinstance(rec.get("celt_alone"), dict):
                e["celt_alone_ms"] = _sig(rec["celt_alone"].get("avg_kernel_ms"), 4)
            par = rec.get("parity") or {}
            e["mismatches"] = sum(int(p.get("mismatches") or 0) for p in par.values() if isinstance(p, dict))
        elif name.endswith("_e2e"):
            e = {"value": _sig(rec.get("samples_per_s_end_to_end")), "cpu": _sig((rec.get("cpu_baseline_e2e") or {}).get("value")),
                 "vs_cpu": _sig(rec.get("vs_cpu_baseline_e2e"), 4), "ok": rec.get("all_ok")}
            if rec.get("samples_per_s_at_the_cpu_quota"):
                e["at_cpu_quota"] = _sig(rec["samples_per_s_at_the_cpu_quota"], 4)       # what the host parse alone allows on this box's CPU quota
                e["host_cpus"] = _sig(rec.get("host_cpus_busy"), 3)
            par = rec.get("parity") or {}
            blocks = list(par.values()) if par and all(isinstance(v, dict) for v in par.values()) else [par]
            e["mismatches"] = sum(int(b.get("mismatches") or 0) for b in blocks)
        elif name == "device_inclusive":
            e = {c: _sig(v.get("samples_per_s_device_inclusive")) for c, v in rec.items() if isinstance(v, dict)}
        elif name in ("vorbis_shapes", "flac_shapes"):
            e = {"shapes": [[str(s["label"]).split(" ")[0] + ("/i32" if s.get("int16_rows") is False else "") if s.get("label")
                             else f"{s.get('channels')}ch/{s.get('blocksize_0')}/{s.get('blocksize_1')}", _sig(s.get("avg_kernel_ms"), 4),
                             _sig(s.get("frac"), 3)] for s in rec.get("shapes", [])], "cols": ["shape", "kernel_ms", "frac"]}
            bad = [s for s in rec.get("shapes", []) if s.get("mismatches") or (s.get("rms_error") is not None and not s["rms_error"] <= 1e-5)]
            e["parity_failures"] = len(bad)
        elif name == "c234_side_by_side":
            e = _pick(rec, ("ms_per_step", "value", "frac_of_peak"))
            e["mismatches"] = sum(int(m or 0) for m in (rec.get("mismatches_by_codec") or {}).values())
        else:
            e = _pick(rec, ("avg_kernel_ms", "frac", "samples_per_s", "mismatches", "bitwise_mismatches", "rms_vs_oracle", "int16_flip_rate"))
            if isinstance(rec.get("parity"), dict) and "mismatches" in rec["parity"]:
                e["mismatches"] = rec["parity"]["mismatches"]
        if rec.get("error"):
            e["error"] = str(rec["error"])[:120]
        others[name] = e
    if others:
        out["other_workloads"] = others
    if full_record_path:
        out["full_record"] = full_record_path
    # the bound is a contract: shed optional detail, least important first, until the line fits
    shed = [("other_workloads", "vorbis_shapes"), ("other_workloads", "flac_shapes"), ("other_workloads", "device_inclusive"),
            ("other_workloads", None), ("roofline", "whole_step"), ("roofline", "mp3_full_fetch"), ("roofline", "vorbis_full_fetch")]
    text = json.dumps(out)
    while len(text) > COMPACT_LIMIT and shed:
        top, sub = shed.pop(0)
        if top in out:
            if sub is None:
                out.pop(top)
            elif isinstance(out[top], dict):
                out[top].pop(sub, None)
        text = json.dumps(out)
    if len(text) > COMPACT_LIMIT:
        out["config"]["workload"] = out["config"].get("workload", "")[:120]
        out["roofline"]["kernels"] = out["roofline"]["kernels"][:4]
        text = json.dumps(out)
    return text


def emit(args, full):
    """Full record -> file (and stdout for `--full-line`); compact line -> the last (normally the only) stdout line."""
    if args.full_line:
        print(json.dumps(full), flush=True)
        return
    path = args.full_record or os.path.join(ROOT, "gpurun_out", "bench_full.json")
    rel = None
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fh:
            json.dump(full, fh, indent=1)
        rel = os.path.relpath(path, ROOT)
    except OSError as e:
        sys.stderr.write(f"bench.py: full record not written ({e})\n")
    print(compact_line(full, rel), flush=True)


# ----------------------------------------------------------------------------------------------------------------
# rank process
# ----------------------------------------------------------------------------------------------------------------
def load_traffic(name):
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as fh:
            return json.load(fh)
    except Exception:
        return None


def run_rank(args, world, rank, local_rank):
    import numpy as np
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)            # barrier + MAX-reduce of a timer only
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev:
        if not args.oversubscribe:
            raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank} but only {ndev} device(s) are visible")
        local_rank %= ndev
    import afgpu
    from afgpu import corpus
    afgpu.lib()
    afgpu.set_device(local_rank)                     # the C library's device for this (the only) host thread
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.current_stream()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def reduce_max(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def reduce_sum(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def timed_steps(wl, steps, warmup):
        """W warm-up + K timed steps of one resident workload; returns (elapsed s, per-part ms lists)."""
        for _ in range(warmup):
            wl.step(stream, None, side)
        barrier()
        ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in wl.parts] for _ in range(steps)]
        t0 = time.perf_counter()
        for i in range(steps):
            wl.step(stream, ev[i], side)                 # events on the stream each kernel is launched on
        barrier()
        elapsed = time.perf_counter() - t0
        per_part = [[ev[i][k][0].elapsed_time(ev[i][k][1]) for i in range(steps)] for k in range(len(wl.parts))]
        return elapsed, per_part

    def reduce_min(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return float(t.item())

    def measure_c5(steps, warmup, want_cpu):
        """BASELINE configs[4] on this rank's shard: (elapsed s of this rank, samples of this rank, per-codec kernel records,
        parity of rank 0's first wave, cpu baseline or None, extras, number of waves, the manifest)."""
        kern5, parity5, cpu5, extra5 = {}, {}, None, {}
        man = corpus.c5_manifest(args.c5_files)
        waves = corpus.c5_shard_waves(man, rank, world, args.c5_wave_files or corpus.C5_WAVE_FILES)
        elapsed5, my5 = 0.0, 0
        for wi, ids in enumerate(waves):
            wl = corpus.build_c5_wave(man, ids, dev)
            if args.only:
                wl.parts = [p for p in wl.parts if p.name in args.only.split(",")]
            e, per_part = timed_steps(wl, steps, warmup)
            elapsed5 += e
            my5 += wl.samples
            for p, ms in zip(wl.parts, per_part):
                k = kern5.setdefault(p.name, {"kernel": p.kernel, "ms": [0.0] * steps, "samples": 0, "alg_bytes": 0, "units": 0})
                k["ms"] = [a + b for a, b in zip(k["ms"], ms)]
                k["samples"] += p.samples; k["alg_bytes"] += p.alg_bytes; k["units"] += p.units
            # the Opus members on their own (in the step they run on a second stream beside the other codecs' kernels, so
            # their event span there is the overlapped one)
            for p in wl.parts:
                if p.name == "celt":
                    ms = []
                    for i in range(3):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(stream); p.launch(stream); e1.record(stream)
                        torch.cuda.synchronize()
                        if i:
                            ms.append(e0.elapsed_time(e1))
                    extra5["celt_alone_ms"] = extra5.get("celt_alone_ms", 0.0) + sum(ms) / len(ms)
                    extra5["celt_alone_bytes"] = extra5.get("celt_alone_bytes", 0) + p.alg_bytes
            if rank == 0 and wi == 0:
                import oraclelib
                parity5 = {p.name: p.check(oraclelib) for p in wl.parts}
                if want_cpu:
                    cpu5 = cpu_baseline(wl.parts, args.cpu_seconds)
            del wl
            torch.cuda.empty_cache()
        # ranks with fewer waves still meet the others' barriers: every rank has the same wave count by construction
        return elapsed5, my5, kern5, parity5, cpu5, extra5, waves, man

    side = torch.cuda.Stream(device=dev)             # the C5 waves put their Opus members on a second stream
    kern = {}                                         # name -> dict(ms list, samples, alg_bytes, units)
    parity, cpu, extra = {}, None, {}
    if args.config == "c5":
        elapsed, my_samples, kern, parity, cpu, extra, waves, man = measure_c5(args.steps, args.warmup, not args.no_cpu_baseline)
        total_samples = reduce_sum(float(my_samples))
        scaling = "strong"
        workload = (f"{args.c5_files}-file mixed corpus (40% MP3 / 25% Ogg Vorbis / 25% FLAC / 10% Opus-CELT, durations "
                    f"log-uniform 4-30 s, seed {corpus.C5_SEED:#x}) file-sharded over {world} GPU(s) by LPT on predicted device time, "
                    f"{len(waves)} wave(s) of <= {args.c5_wave_files or corpus.C5_WAVE_FILES} files per GPU, device-resident records -> PCM")
        cfg_extra = {"files": args.c5_files, "waves_per_gpu": len(waves), "files_this_gpu": int(sum(len(w) for w in waves)),
                     "lpt_imbalance": corpus.c5_imbalance(man, world),
                     "partition": "LPT on predicted device time (samples x measured ns/sample per codec), the 64 longest Opus files dealt out first"}
    else:
        which = {"c234": ("mp3", "vorbis", "flac"), "c2": ("mp3",), "c3": ("vorbis",), "c4": ("flac",)}[args.config]
        wl = corpus.build_c234(dev, rank, which, args.files, args.seg)
        elapsed, per_part = timed_steps(wl, args.steps, args.warmup)
        my_samples = wl.samples
        for p, ms in zip(wl.parts, per_part):
            kern[p.name] = {"kernel": p.kernel, "ms": ms, "samples": p.samples, "alg_bytes": p.alg_bytes, "units": p.units,
                            "survey_bytes": p.survey_bytes}
        total_samples = float(my_samples) * world
        scaling = "weak"
        names = {"mp3": f"{args.files} x MP3 CBR-128k stereo 60 s (C2)", "vorbis": f"{args.files} x Ogg Vorbis 2048/256 stereo, 2584 packets (C3)",
                 "flac": f"{4 * args.files} x FLAC 16-bit stereo, 323 frames of 4096, LPC order 8/12 (C4"
                         + ("; residual rows int32" if os.environ.get("AFG_FLAC_RES32") else "; residual rows int16, as the host parser packs 16-bit material") + ")"}
        workload = " + ".join(names[w] for w in which) + " per GPU, resident together; device-resident transform-stage records -> PCM"
        cfg_extra = {"files_per_gpu": {w: (4 * args.files if w == "flac" else args.files) for w in which}}
        if rank == 0:
            import oraclelib
            parity = {p.name: p.check(oraclelib, 2) for p in wl.parts}
            # MP3 with every subband fetched (no AFG_MP3_NZ_BANDS declaration): the same kernel moving all algorithmic bytes
            mp3 = next((p for p in wl.parts if p.name == "mp3"), None)
            if mp3 is not None and not args.no_full_fetch:
                ms = []
                for i in range(2 + min(args.steps, 10)):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream); mp3.launch(stream, full_fetch=True); e1.record(stream)
                    torch.cuda.synchronize()
                    if i >= 2:
                        ms.append(e0.elapsed_time(e1))
                extra["mp3_full_fetch"] = {"avg_kernel_ms": sum(ms) / len(ms), "frac": mp3.alg_bytes / (sum(ms) / len(ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                           "note": "flag words without AFG_MP3_NZ_BANDS: all 32 subbands of every block fetched"}
            # Vorbis likewise with nothing declared (no AFG_VORBIS_NZ_EIGHTHS: what a file whose residue runs to the top of the
            # band gets): the headline's Vorbis figure owes its last 5 % to a property of the synthetic spectra
            vb = next((p for p in wl.parts if p.name == "vorbis"), None)
            if vb is not None and not args.no_full_fetch:
                ms = []
                for i in range(2 + min(args.steps, 10)):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream); vb.launch(stream, full_fetch=True); e1.record(stream)
                    torch.cuda.synchronize()
                    if i >= 2:
                        ms.append(e0.elapsed_time(e1))
                extra["vorbis_full_fetch"] = {"avg_kernel_ms": sum(ms) / len(ms), "samples_per_s": vb.samples / (sum(ms) / len(ms) * 1e-3),
                                              "frac": vb.survey_bytes / (sum(ms) / len(ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                              "note": "packet flags without AFG_VORBIS_NZ_EIGHTHS: every long block's whole spectrum fetched (40.1 GB, SURVEY 8d's figure)"}
                vb.launch(stream)                          # leave the declared launch's output in the plane
                torch.cuda.synchronize()
            # the same step with its kernels side by side on three streams, the persistent ones (MP3, Vorbis) launched first
            # and FLAC's grid filling in behind them: not the line's `value` -- there every kernel has the device to itself,
            # which is what a per-kernel roofline needs -- but what a caller gets who launches the three batches together
            if len(wl.parts) == 3 and not args.no_others:
                lanes = [torch.cuda.Stream(device=dev) for _ in wl.parts]
                order = sorted(range(len(wl.parts)), key=lambda i: {"vorbis": 0, "mp3": 1}.get(wl.parts[i].name, 2))
                n_sbs = max(3, min(args.steps, 10))
                for _ in range(2):
                    wl.step_side_by_side(stream, lanes, None, None, order)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n_sbs):
                    wl.step_side_by_side(stream, lanes, None, None, order)
                torch.cuda.synchronize()
                sbs = (time.perf_counter() - t0) / n_sbs
                sbs_parity = {p.name: p.check(oraclelib, 1) for p in wl.parts}
                extra["side_by_side"] = {"workload": "the headline step with its three kernels on three streams (launch order " +
                                                     ", ".join(wl.parts[i].name for i in order) + "), joined per step",
                                         "ms_per_step": sbs * 1e3, "value": wl.samples / sbs, "unit": "samples/s", "steps": n_sbs,
                                         "achieved_GBs": wl.alg_bytes / sbs / 1e9, "frac_of_peak": wl.alg_bytes / sbs / 1e9 / HBM_PEAK_GBS,
                                         "mismatches_by_codec": {k: v.get("mismatches") for k, v in sbs_parity.items()},
                                         "note": "per-kernel spans overlap here; `value`, `roofline` and `kernels` above come from the serial step"}
                del lanes
            # device-to-device copy with the library's streaming copy kernel: the copy rate this box sustains for a
            # read-N / write-N stream, reported next to the 8 TB/s spec the roofline is priced against
            nbytes = 2 << 30
            try:
                a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                b = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                cms = []
                for i in range(4):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream); afgpu.copy_probe(b, a, nbytes); e1.record(stream)
                    torch.cuda.synchronize()
                    if i:
                        cms.append(e0.elapsed_time(e1))
                extra["measured_copy_GBs"] = 2 * nbytes / (min(cms) * 1e-3) / 1e9
                del a, b
            except torch.cuda.OutOfMemoryError:
                extra["measured_copy_GBs"] = None
            if not args.no_cpu_baseline:
                cpu = cpu_baseline(wl.parts, args.cpu_seconds)       # rank 0's host cores, at every N

    elapsed = reduce_max(elapsed)
    # ---- N > 1, default config: the scaling configuration itself (BASELINE configs[4], strong scaling) on the same ranks ----
    c5_tail = None
    if world > 1 and args.config == "c234" and not args.no_others:
        wl = mp3 = p = None                              # release the headline batches: the C5 waves need the room
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        barrier()
        e5, my5, kern5, parity5, _, extra5, waves5, man5 = measure_c5(3, 1, False)
        e5_max, e5_min = reduce_max(e5), reduce_min(e5)
        tot5 = reduce_sum(float(my5))
        if rank == 0:
            value5 = tot5 * 3 / e5_max
            ref = (load_traffic("r06_c5.json") or load_traffic("r05_c5.json") or {})
            ref_value = ref.get("value") if ref.get("n_gpus", 1) == 1 else None
            c5_tail = {"workload": f"{args.c5_files}-file mixed corpus file-sharded over {world} GPU(s) by LPT on predicted device time (bench.py --config c5)",
                       "value": value5, "unit": "samples/s", "scaling": "strong", "n_gpus": world, "steps": 3, "ms_per_step": e5_max / 3 * 1e3,
                       "per_rank_ms_per_step": {"min": e5_min / 3 * 1e3, "max": e5_max / 3 * 1e3},
                       "samples_per_step": int(tot5), "waves_per_gpu": len(waves5), "lpt_imbalance": corpus.c5_imbalance(man5, world),
                       "n1_reference_value": ref_value, "n1_reference_source": "profiles/r06_c5.json or r05_c5.json (bench.py --config c5 on one GPU)" if ref_value else None,
                       "efficiency_vs_n1": (value5 / (world * ref_value)) if ref_value else None,
                       "kernels": [{"codec": n, "avg_kernel_ms": sum(k["ms"]) / len(k["ms"]), "samples_per_launch": int(k["samples"])} for n, k in kern5.items()],
                       "parity": parity5}
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return 0

    # ---- the line ----
    steps = args.steps
    value = total_samples * steps / elapsed
    # HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc
    # passes over the same full-size batch, corrected as MI355X_MICROARCH.md prescribes: tools/pmc_collect.sh): only
    # for the kernels and batch sizes those passes were taken on
    pmc_file = {"mp3": "r06_pmc_mp3_tolerance_kernel.json", "vorbis": "r06_pmc_vorbis_walk_kernel.json",
                "flac": "r06_pmc_flac_restore1_kernel.json"}     # (FLAC: counters calibrated on its own access pattern, both instantiations)
    kernels = []
    for name, k in kern.items():
        avg_ms = sum(k["ms"]) / len(k["ms"])
        ach = k["alg_bytes"] / (avg_ms * 1e-3) / 1e9
        tb = None
        # (only for the run those passes describe: full size, the product library, no input-format or segment overrides)
        variant = args.seg or any(os.environ.get(v) for v in ("AFG_LIB_PATH", "AFG_FLAC_RES32", "AFG_MP3_FLOAT_UPLOAD", "AFG_NUMERIC"))
        live = getattr(args, "live_traffic", None) or {}
        if name in live and not variant:
            tb = live[name]
        elif args.config in ("c234", "c2", "c3", "c4") and args.files == 1024 and name in pmc_file and not variant:
            tb = ((load_traffic(pmc_file[name]) or {}).get("derived") or {}).get("hbm_bytes_per_launch")
        kernels.append({"codec": name, "kernel": k["kernel"], "avg_kernel_ms": avg_ms, "achieved": ach, "frac": ach / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_launch": int(k["alg_bytes"]), "units_per_launch": int(k["units"]),
                        "samples_per_launch": int(k["samples"]), "samples_per_s": k["samples"] / (avg_ms * 1e-3),
                        "traffic": tb, "frac_by_traffic": (tb / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if tb else None,
                        "traffic_source": (("measured before this run (--measure-traffic: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py --config "
                                            + {"mp3": "c2", "vorbis": "c3", "flac": "c4"}.get(name, "?") + ")") if name in live
                                           else ("profiles/" + pmc_file[name])) if tb else None})
        if k.get("survey_bytes") and k["survey_bytes"] != k["alg_bytes"]:
            # `frac` above is priced on the bytes this launch has to move; SURVEY 8(d)'s per-unit figure is shown beside it
            kernels[-1]["bytes_at_survey_8d"] = int(k["survey_bytes"])
            kernels[-1]["frac_at_survey_8d"] = k["survey_bytes"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            if name == "flac":
                # int16 residual rows: the launch reads 2 B / sample where SURVEY 8(d) counts 4 (8 B / sample with the store)
                kernels[-1]["launch"] = ("the populated instantiations of flac_restore1_kernel (here LPC order <= 8 and <= 12) run side by side on two "
                                         "streams: a kernel trace lists each with about this duration, and they overlap")
                kernels[-1]["input_rows"] = "int16 residual rows (6 B / sample moved)"
            elif name == "vorbis":
                kernels[-1]["input_rows"] = ("packet flags declare the non-zero eighths of each long block's spectrum (AFG_VORBIS_NZ_EIGHTHS): "
                                             "the zero tail above the residue's end is not fetched")
            elif name == "mp3":
                kernels[-1]["input_rows"] = "flag words declare the non-zero subbands of each block (AFG_MP3_NZ_BANDS): the zero tail is not fetched"
    dom = max(kernels, key=lambda d: d["avg_kernel_ms"])
    step_ms = sum(d["avg_kernel_ms"] for d in kernels)
    step_bytes = sum(d["algorithmic_bytes_per_launch"] for d in kernels)
    roofline = {"bound": "hbm", "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"],
                "traffic": dom["traffic"], "traffic_from": dom["traffic_source"], "frac_by_traffic": dom["frac_by_traffic"],
                "kernel": dom["kernel"], "codec": dom["codec"], "avg_kernel_ms": dom["avg_kernel_ms"],
                "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
                "note": "dominant kernel = the codec with the largest share of a step's time; every kernel of the step is listed in `kernels`",
                "kernels": kernels,
                "overlapped_on_a_second_stream": ["celt"] if args.config == "c5" else [],
                "whole_step": {"algorithmic_bytes": int(step_bytes), "kernel_ms": step_ms,
                               "achieved": step_bytes / (step_ms * 1e-3) / 1e9, "frac": step_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "measured_copy_GBs": extra.get("measured_copy_GBs"), "mp3_full_fetch": extra.get("mp3_full_fetch"),
                "vorbis_full_fetch": extra.get("vorbis_full_fetch")}
    if extra.get("celt_alone_ms"):
        roofline["celt_alone"] = {"avg_kernel_ms": extra["celt_alone_ms"], "frac": extra["celt_alone_bytes"] / (extra["celt_alone_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "note": "the Opus members of all waves launched on their own (no other kernel beside them), summed over the waves"}
    line = {
        "metric": METRIC, "value": value, "unit": "samples/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "f32 (MP3, Vorbis, CELT) + int32 (FLAC)", "data": "synthetic",
        "config": dict({"workload": workload, "name": args.config, "samples_per_step": int(total_samples),
                        "parallelism": f"file-sharded x{world}, no collective on the data path (gloo barrier + MAX of the timer only)"},
                       **cfg_extra),
        "roofline": roofline, "cpu_baseline": cpu, "parity": parity,
    }
    if args.oversubscribe and world > ndev:
        line["oversubscribed"] = f"{world} ranks on {ndev} device(s): a launcher test, not a scaling point"
    failed = [n for n, p in parity.items() if p["mismatches"]]
    if world == 1 and args.config == "c234" and not args.no_others:
        # release the headline batches (148 GB) first: the C5 waves need the room
        wl = mp3 = p = None
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        line["other_workloads"] = other_workloads(args)
        failed += other_parity_failures(line["other_workloads"])
    if c5_tail is not None:
        line["other_workloads"] = {"c5": c5_tail}
        failed += other_parity_failures(line["other_workloads"])
    if extra.get("side_by_side"):
        line.setdefault("other_workloads", {})["c234_side_by_side"] = extra["side_by_side"]
        failed += [f"side_by_side:{n}" for n, m in extra["side_by_side"]["mismatches_by_codec"].items() if m]
    emit(args, line)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        sys.stderr.write(f"bench.py: parity failure against the oracle in {failed}\n")
        return 1
    return 0


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None:
        if args.gpus > 1:
            return launch_ranks(args)                  # before anything in this process touches a GPU
        world, rank, local_rank = 1, 0, 0
        # (by default only in the full default run -- the one the driver times -- and never inside this script's own children)
        want = args.measure_traffic if args.measure_traffic is not None else (not args.no_others and not os.environ.get("AFG_BENCH_NESTED"))
        if want and args.config == "c234" and args.files == 1024:
            args.live_traffic = measure_traffic()      # child processes: this one has not touched a GPU yet
    else:
        world = int(env_world)
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
        if args.gpus != world:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; refusing to report a mislabelled run\n")
            return 2
    return run_rank(args, world, rank, local_rank)


if __name__ == "__main__":
    sys.exit(main())
