"""ctypes view of oracle/liboracle_afg.so (the CPU restatement; test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle_afg.so")


def build(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")


class VorbisTables(C.Structure):
    _fields_ = [("n", C.c_int), ("A", C.POINTER(C.c_float)), ("B", C.POINTER(C.c_float)),
                ("C", C.POINTER(C.c_float)), ("window", C.POINTER(C.c_float)),
                ("bitrev", C.POINTER(C.c_uint16))]


FLAC_SUBFRAME_DTYPE = np.dtype([("coef", np.int16, (32,)), ("order", np.uint8), ("shift", np.uint8),
                                ("wasted", np.uint8), ("use64", np.uint8)], align=True)
FLAC_FRAME_DTYPE = np.dtype([("in_off", np.uint64), ("out_off", np.uint64), ("block_size", np.uint32),
                             ("sf_index", np.uint32), ("channels", np.uint8), ("assignment", np.uint8),
                             ("bps", np.uint8), ("res16", np.uint8), ("pad", np.uint8, (4,))], align=True)
assert FLAC_SUBFRAME_DTYPE.itemsize == 68 and FLAC_FRAME_DTYPE.itemsize == 32

CELT_FRAME_DTYPE = np.dtype([("coef_off", np.uint64), ("out_off", np.uint64), ("out_stride", np.uint32),
                             ("frame_size", np.uint16), ("blocks", np.uint8), ("pad", np.uint8),
                             ("pf_period_new", np.int32), ("pf_gains_new", np.float32, (3,)),
                             ("imdct_scale", np.float32), ("pad2", np.uint32)], align=True)
assert CELT_FRAME_DTYPE.itemsize == 48
CELT_STATE_FLOATS = 2064
QOA_FRAME_DTYPE = np.dtype([("byte_off", np.uint64), ("out_off", np.uint64), ("samples", np.uint16),
                            ("channels", np.uint8), ("pad", np.uint8, (5,))], align=True)
assert QOA_FRAME_DTYPE.itemsize == 24

MP3_STATE_FLOATS = 2 * 288 + 960


def mp3_flags(block_type=0, n_long_bands=0, aa_bands=31):
    return np.uint32(block_type | (n_long_bands << 8) | ((aa_bands + 1) << 16))


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    L.afgo_mp3_antialias.argtypes = [f32p, C.c_int]
    L.afgo_mp3_imdct_gr.argtypes = [f32p, f32p, C.c_uint, C.c_uint]
    L.afgo_mp3_change_sign.argtypes = [f32p]
    L.afgo_mp3_dct2.argtypes = [f32p, C.c_int]
    L.afgo_mp3_synth_granule.argtypes = [f32p, f32p, C.c_int, C.c_int, f32p, f32p]
    L.afgo_mp3_transform.argtypes = [C.c_uint32, u32p, u8p, f32p, u32p, f32p, C.c_void_p]
    for fn in (L.afgo_mp3_antialias, L.afgo_mp3_imdct_gr, L.afgo_mp3_change_sign, L.afgo_mp3_dct2,
               L.afgo_mp3_synth_granule, L.afgo_mp3_transform):
        fn.restype = None
    L.afgo_vorbis_tables_init.argtypes = [C.POINTER(VorbisTables), C.c_int]
    L.afgo_vorbis_tables_init.restype = C.c_int
    L.afgo_vorbis_tables_free.argtypes = [C.POINTER(VorbisTables)]
    L.afgo_vorbis_tables_free.restype = None
    L.afgo_vorbis_inverse_mdct.argtypes = [f32p, C.c_int, C.POINTER(VorbisTables), f32p]
    L.afgo_vorbis_inverse_mdct.restype = None
    L.afgo_vorbis_layout.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int, u8p, C.c_uint64, C.c_uint64,
                                     u64p, u64p, C.POINTER(C.c_uint64)]
    L.afgo_vorbis_layout.restype = C.c_uint64
    L.afgo_vorbis_transform.argtypes = [C.c_uint32, u32p, u8p, u16p, u16p, u8p, u64p, u64p, f32p, f32p]
    L.afgo_vorbis_transform.restype = C.c_int
    L.afgo_flac_prediction_32.argtypes = [C.c_uint, C.c_int, C.c_void_p, C.c_void_p]
    L.afgo_flac_prediction_32.restype = C.c_int32
    L.afgo_flac_prediction_64.argtypes = [C.c_uint, C.c_int, C.c_void_p, C.c_void_p]
    L.afgo_flac_prediction_64.restype = C.c_int32
    L.afgo_flac_transform.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, i32p, i32p, C.c_void_p]
    L.afgo_flac_transform.restype = None
    L.afgo_celt_imdct_half.argtypes = [C.c_int, f32p, f32p, C.c_int, C.c_float]
    L.afgo_celt_imdct_half.restype = None
    L.afgo_celt_transform.argtypes = [C.c_uint32, u64p, C.c_void_p, f32p, f32p, C.c_void_p]
    L.afgo_celt_transform.restype = None
    L.afgo_opus_output.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.afgo_opus_output.restype = None
    L.afgo_qoa_transform.argtypes = [C.c_uint64, C.c_void_p, u8p, C.c_void_p, C.c_void_p]
    L.afgo_qoa_transform.restype = None
    L.afgo_qoa_encode.argtypes = [i16p, C.c_uint32, C.c_int, C.c_uint32, u8p, C.c_void_p]
    L.afgo_qoa_encode.restype = C.c_size_t
    _lib = L
    return L


# ---------------------------------------------------------------- MP3 ------
def mp3_transform(ngr, nch, coef, flags, want_state=False):
    """Whole-batch MP3 transform stage on the CPU oracle.  Returns pcm (and states)."""
    ngr = np.ascontiguousarray(ngr, np.uint32)
    nch = np.ascontiguousarray(nch, np.uint8)
    coef = np.ascontiguousarray(coef, np.float32).reshape(-1)
    flags = np.ascontiguousarray(flags, np.uint32)
    nblk = int((ngr.astype(np.int64) * nch).sum())
    assert coef.size == nblk * 576 and flags.size == nblk
    pcm = np.zeros(nblk * 576, np.float32)
    states = np.zeros((len(ngr), MP3_STATE_FLOATS), np.float32) if want_state else None
    lib().afgo_mp3_transform(len(ngr), ngr, nch, coef, flags, pcm,
                             states.ctypes.data if want_state else None)
    return (pcm, states) if want_state else pcm


def mp3_transform_into(ngr, nch, coef, flags, pcm):
    """Same as mp3_transform but into a caller-owned buffer (no allocation: used when timing)."""
    lib().afgo_mp3_transform(len(ngr), ngr, nch, coef, flags, pcm, None)
    return pcm


# ------------------------------------------------------------- Vorbis ------
def vorbis_tables(n):
    t = VorbisTables()
    assert lib().afgo_vorbis_tables_init(C.byref(t), n) == 0
    out = dict(
        A=np.ctypeslib.as_array(t.A, (n // 2,)).copy(), B=np.ctypeslib.as_array(t.B, (n // 2,)).copy(),
        C=np.ctypeslib.as_array(t.C, (n // 4,)).copy(), window=np.ctypeslib.as_array(t.window, (n // 2,)).copy(),
        bitrev=np.ctypeslib.as_array(t.bitrev, (n // 8,)).copy())
    lib().afgo_vorbis_tables_free(C.byref(t))
    return out


def vorbis_inverse_mdct(spec, n):
    t = VorbisTables()
    assert lib().afgo_vorbis_tables_init(C.byref(t), n) == 0
    buf = np.zeros(n, np.float32)
    buf[: n // 2] = spec
    scratch = np.zeros(n // 2, np.float32)
    lib().afgo_vorbis_inverse_mdct(buf, n, C.byref(t), scratch)
    lib().afgo_vorbis_tables_free(C.byref(t))
    return buf


def vorbis_layout(npkt, nch, bs0, bs1, pflags):
    """Batch-wide spec/out offsets (in floats) for concatenated streams."""
    npkt = np.ascontiguousarray(npkt, np.uint32)
    pflags = np.ascontiguousarray(pflags, np.uint8)
    tot = int(npkt.sum())
    spec_off = np.zeros(tot, np.uint64)
    out_off = np.zeros(tot, np.uint64)
    sb, ob, p0 = 0, 0, 0
    for s in range(len(npkt)):
        st = C.c_uint64(0)
        k = int(npkt[s])
        so = np.zeros(k, np.uint64)
        oo = np.zeros(k, np.uint64)
        ob = lib().afgo_vorbis_layout(k, int(nch[s]), int(bs0[s]), int(bs1[s]),
                                      np.ascontiguousarray(pflags[p0:p0 + k]), sb, ob, so, oo, C.byref(st))
        spec_off[p0:p0 + k] = so
        out_off[p0:p0 + k] = oo
        sb = st.value
        p0 += k
    return spec_off, out_off, sb, ob


def vorbis_transform(npkt, nch, bs0, bs1, pflags, spec_off, out_off, spec, out_total):
    out = np.zeros(int(out_total), np.float32)
    rc = lib().afgo_vorbis_transform(
        len(npkt), np.ascontiguousarray(npkt, np.uint32), np.ascontiguousarray(nch, np.uint8),
        np.ascontiguousarray(bs0, np.uint16), np.ascontiguousarray(bs1, np.uint16),
        np.ascontiguousarray(pflags, np.uint8), np.ascontiguousarray(spec_off, np.uint64),
        np.ascontiguousarray(out_off, np.uint64), np.ascontiguousarray(spec, np.float32), out)
    assert rc == 0, rc
    return out


def vorbis_floor(packets, curves, points, steps, spec):
    """afgo_vorbis_floor: inverse coupling + floor curves on residue vectors; records in the product's dtypes.  Returns the spectra."""
    out = np.ascontiguousarray(spec, np.float32).copy()
    packets = np.ascontiguousarray(packets)
    curves = np.ascontiguousarray(curves)
    assert packets.dtype.itemsize == 32 and curves.dtype.itemsize == 8
    points = np.ascontiguousarray(points, np.int32)
    steps = np.ascontiguousarray(steps, np.uint8)
    fn = lib().afgo_vorbis_floor
    fn.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    fn.restype = None
    fn(len(packets), packets.ctypes.data, curves.ctypes.data, points.ctypes.data, steps.ctypes.data, out.ctypes.data)
    return out


class _Mp3File(C.Structure):
    _fields_ = [("channels", C.c_int), ("hz", C.c_int), ("vbr_tag_found", C.c_int), ("start_delay", C.c_int),
                ("detected_samples", C.c_uint64), ("samples", C.c_uint64), ("n_streams", C.c_uint32),
                ("stream_granules", C.POINTER(C.c_uint32)), ("n_blocks", C.c_uint64), ("coef", C.POINTER(C.c_float)),
                ("flags", C.POINTER(C.c_uint32)), ("pcm_samples", C.c_uint64), ("pcm", C.POINTER(C.c_float)), ("layer", C.c_int)]


def mp3_decode_file(data):
    """Oracle front-end + transform over a whole file in memory.  Returns None if no Layer III stream is found,
    else a dict: channels, hz, tagged, start_delay, detected_samples, declared_samples, runs, coef, flags, pcm."""
    buf = bytes(data)
    f = _Mp3File()
    fn = lib().afgo_mp3_decode_file
    fn.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(_Mp3File)]
    rc = fn(buf, len(buf), C.byref(f))
    if rc != 0:
        return None
    try:
        def arr(ptr, n, shape=None):
            a = np.ctypeslib.as_array(ptr, shape=(n,)).copy() if n else np.zeros(0, np.float32)
            return a.reshape(shape) if shape else a
        nb = int(f.n_blocks)
        return {"channels": f.channels, "hz": f.hz, "layer": f.layer, "tagged": f.vbr_tag_found, "start_delay": f.start_delay,
                "detected_samples": int(f.detected_samples), "declared_samples": int(f.samples),
                "runs": arr(f.stream_granules, int(f.n_streams)).astype(np.uint32) if f.n_streams else np.zeros(0, np.uint32),
                "coef": arr(f.coef, nb * 576, (-1, 576)) if nb else np.zeros((0, 576), np.float32),
                "flags": arr(f.flags, nb).astype(np.uint32) if nb else np.zeros(0, np.uint32),
                "pcm": arr(f.pcm, int(f.pcm_samples))}
    finally:
        free = lib().afgo_mp3_file_free
        free.argtypes = [C.POINTER(_Mp3File)]
        free.restype = None
        free(C.byref(f))


class _VorbisFile(C.Structure):
    _fields_ = [("channels", C.c_int), ("sample_rate", C.c_uint), ("blocksize0", C.c_int), ("blocksize1", C.c_int),
                ("total_samples", C.c_uint32), ("n_packets", C.c_uint32), ("pflags", C.POINTER(C.c_uint8)),
                ("spec_floats", C.c_uint64), ("spec", C.POINTER(C.c_float)), ("take_from", C.POINTER(C.c_int32)),
                ("take_count", C.POINTER(C.c_int32)), ("pcm_frames", C.c_uint64)]


def vorbis_decode_file(data, seek_clears_eof=False):
    """Oracle Ogg Vorbis front-end over a whole file in memory (records up to the transform seam + delivery plan).
    None if the data is not an Ogg Vorbis stream the reference accepts.  seek_clears_eof=True: upstream stb_vorbis'
    seek (see afgo_vorbis_decode_file_ex)."""
    buf = bytes(data)
    f = _VorbisFile()
    fn = lib().afgo_vorbis_decode_file_ex
    fn.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(_VorbisFile), C.c_int]
    if fn(buf, len(buf), C.byref(f), int(seek_clears_eof)) != 0:
        return None
    try:
        n = int(f.n_packets)

        def arr(ptr, cnt, dtype):
            return np.ctypeslib.as_array(ptr, shape=(cnt,)).astype(dtype) if cnt else np.zeros(0, dtype)
        return {"channels": f.channels, "sample_rate": f.sample_rate, "blocksize0": f.blocksize0, "blocksize1": f.blocksize1,
                "total_samples": int(f.total_samples), "pflags": arr(f.pflags, n, np.uint8),
                "spec": arr(f.spec, int(f.spec_floats), np.float32), "take_from": arr(f.take_from, n, np.int32),
                "take_count": arr(f.take_count, n, np.int32), "pcm_frames": int(f.pcm_frames)}
    finally:
        free = lib().afgo_vorbis_file_free
        free.argtypes = [C.POINTER(_VorbisFile)]
        free.restype = None
        free(C.byref(f))


def vorbis_file_pcm(rec):
    """records of vorbis_decode_file -> transform oracle -> the frames the pull API delivers ([frames, channels])."""
    n, ch = len(rec["pflags"]), rec["channels"]
    if n == 0:
        return np.zeros((0, ch), np.float32)
    so, oo, _, total = vorbis_layout(np.array([n], np.uint32), [ch], [rec["blocksize0"]], [rec["blocksize1"]], rec["pflags"])
    out = vorbis_transform(np.array([n], np.uint32), np.array([ch], np.uint8), np.array([rec["blocksize0"]], np.uint16),
                           np.array([rec["blocksize1"]], np.uint16), rec["pflags"], so, oo, rec["spec"], total)
    parts = [out[int(oo[p]) + int(rec["take_from"][p]) * ch: int(oo[p]) + (int(rec["take_from"][p]) + int(rec["take_count"][p])) * ch]
             for p in range(n) if rec["take_count"][p] > 0]
    return (np.concatenate(parts) if parts else np.zeros(0, np.float32)).reshape(-1, ch)


# --------------------------------------------------------------- FLAC ------
def flac_transform(frames, subframes, res, out_total, want_float=False):
    frames = np.ascontiguousarray(frames, FLAC_FRAME_DTYPE)
    subframes = np.ascontiguousarray(subframes, FLAC_SUBFRAME_DTYPE)
    res = np.ascontiguousarray(res, np.int32)
    out = np.zeros(int(out_total), np.int32)
    outf = np.zeros(int(out_total), np.float32) if want_float else None
    lib().afgo_flac_transform(len(frames), frames.ctypes.data, subframes.ctypes.data, res, out,
                              outf.ctypes.data if want_float else None)
    return (out, outf) if want_float else out


# ---------------------------------------------------------------- QOA ------
def qoa_encode(pcm, samplerate=44100):
    """pcm: int16 [frames, channels].  Returns (file bytes, the encoder's own reconstruction)."""
    pcm = np.ascontiguousarray(pcm, np.int16)
    n, ch = pcm.shape
    nfr = (n + 5119) // 5120
    out = np.zeros(8 + nfr * (8 + 16 * ch + 8 * 256 * ch) + 64, np.uint8)
    recon = np.zeros_like(pcm)
    size = lib().afgo_qoa_encode(pcm.reshape(-1), n, ch, samplerate, out, recon.ctypes.data)
    return out[:size].copy(), recon


def qoa_transform(frames, data, out_total, want_float=True):
    frames = np.ascontiguousarray(frames, QOA_FRAME_DTYPE)
    data = np.ascontiguousarray(data, np.uint8)
    oi = np.zeros(int(out_total), np.int16)
    of = np.zeros(int(out_total), np.float32) if want_float else None
    lib().afgo_qoa_transform(len(frames), frames.ctypes.data, data, oi.ctypes.data,
                             of.ctypes.data if want_float else None)
    return (oi, of) if want_float else oi


# --------------------------------------------------------------- CELT ------
def opus_output(x):
    """float32 array -> (int16, float32) as OpusFile.readFrame + stream.d:480 deliver them."""
    x = np.ascontiguousarray(x, np.float32).reshape(-1)
    oi = np.zeros(x.size, np.int16)
    of = np.zeros(x.size, np.float32)
    lib().afgo_opus_output(x.size, x.ctypes.data, oi.ctypes.data, of.ctypes.data)
    return oi, of


def celt_transform(rec_base, recs, coeffs, out_total, states=None):
    rec_base = np.ascontiguousarray(rec_base, np.uint64)
    recs = np.ascontiguousarray(recs, CELT_FRAME_DTYPE)
    out = np.zeros(int(out_total), np.float32)
    lib().afgo_celt_transform(len(rec_base) - 1, rec_base, recs.ctypes.data,
                              np.ascontiguousarray(coeffs, np.float32), out,
                              states.ctypes.data if states is not None else None)
    return out


class _OpusFile(C.Structure):
    _fields_ = [("channels", C.c_int32), ("preskip", C.c_int32), ("gain_i", C.c_int32), ("error", C.c_int32),
                ("gain", C.c_float), ("pad", C.c_int32), ("declared_frames", C.c_int64), ("pcm_frames", C.c_uint64),
                ("n_frames", C.c_uint64), ("frames", C.c_void_p), ("n_coeffs", C.c_uint64), ("coeffs", C.POINTER(C.c_float))]


def opus_decode_file(data):
    """Oracle Ogg Opus front-end (CELT-only) over a whole file in memory: transform-stage records of channel 0 per
    frame + the coefficient plane (channels consecutive per frame).  Returns the integer status (-1 not an Opus file the
    reference opens, -3 SILK / hybrid) instead of a dict on failure."""
    buf = bytes(data)
    f = _OpusFile()
    fn = lib().afgo_opus_decode_file
    fn.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(_OpusFile)]
    fn.restype = C.c_int
    rc = fn(buf, len(buf), C.byref(f))
    if rc != 0:
        return rc
    try:
        n = int(f.n_frames)
        frames = np.frombuffer(C.string_at(f.frames, n * CELT_FRAME_DTYPE.itemsize), CELT_FRAME_DTYPE).copy() if n \
            else np.zeros(0, CELT_FRAME_DTYPE)
        nc = int(f.n_coeffs)
        coeffs = np.ctypeslib.as_array(f.coeffs, shape=(nc,)).copy() if nc else np.zeros(0, np.float32)
        return {"channels": f.channels, "preskip": f.preskip, "gain_i": f.gain_i, "gain": float(np.float32(f.gain)),
                "error": bool(f.error), "declared_frames": int(f.declared_frames), "pcm_frames": int(f.pcm_frames),
                "frames": frames, "coeffs": coeffs}
    finally:
        free = lib().afgo_opus_file_free
        free.argtypes = [C.POINTER(_OpusFile)]
        free.restype = None
        free(C.byref(f))


def opus_channel_records(rec):
    """records of opus_decode_file -> (rec_base, recs) of afgo_celt_transform: one sequence per output channel."""
    ch, fr = rec["channels"], rec["frames"]
    recs = np.zeros(ch * len(fr), CELT_FRAME_DTYPE)
    for c in range(ch):
        part = fr.copy()
        part["coef_off"] += np.uint64(c) * part["frame_size"].astype(np.uint64)
        part["out_off"] += np.uint64(c)
        recs[c * len(fr):(c + 1) * len(fr)] = part
    return np.arange(ch + 1, dtype=np.uint64) * np.uint64(len(fr)), recs


def opus_file_pcm(rec):
    """records of opus_decode_file -> what AudioStream.readSamplesFloat delivers: transform oracle, gain (dopus.d:6690),
    int16 conversion / 32767 (dopus.d:8098-8105, stream.d:480), clipped to the declared length (stream.d:439-442)."""
    ch = rec["channels"]
    base, recs = opus_channel_records(rec)
    pcm = celt_transform(base, recs, rec["coeffs"], rec["pcm_frames"] * ch)
    if rec["gain_i"]:
        pcm = pcm * np.float32(rec["gain"])
    _, of = opus_output(pcm)
    return of.reshape(-1, ch)[:max(0, min(rec["pcm_frames"], rec["declared_frames"]))]


class _FlacFile(C.Structure):
    _fields_ = [("channels", C.c_uint32), ("sample_rate", C.c_uint32), ("bps", C.c_uint32), ("max_block", C.c_uint32),
                ("total_samples", C.c_uint64), ("n_samples", C.c_uint64), ("pcm", C.POINTER(C.c_int32)),
                ("n_frames", C.c_uint32), ("flags", C.c_uint32), ("first_flag_sample", C.c_uint64)]


FLAC_F_IGNORED_FAILURE, FLAC_F_UNINITIALISED, FLAC_F_UNDEFINED = 1, 2, 4


def flac_decode_file(data):
    """Oracle FLAC front-end (oracle/flac_frontend.c) over a whole native FLAC file in memory: what drflac_read_s32
    delivers to the end of the stream.  Returns the integer status (-1: not a file the reference opens) on failure."""
    buf = bytes(data)
    f = _FlacFile()
    fn = lib().afgo_flac_decode_file
    fn.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(_FlacFile)]
    fn.restype = C.c_int
    rc = fn(buf, len(buf), C.byref(f))
    if rc != 0:
        return rc
    try:
        n = int(f.n_samples)
        pcm = np.ctypeslib.as_array(f.pcm, shape=(n,)).copy() if n else np.zeros(0, np.int32)
        first = int(f.first_flag_sample)
        return {"channels": int(f.channels), "sample_rate": int(f.sample_rate), "bps": int(f.bps), "max_block": int(f.max_block),
                "total_samples": int(f.total_samples), "pcm": pcm, "n_frames": int(f.n_frames), "flags": int(f.flags),
                "first_flag_sample": None if first == 2 ** 64 - 1 else first}
    finally:
        free = lib().afgo_flac_file_free
        free.argtypes = [C.POINTER(_FlacFile)]
        free.restype = None
        free(C.byref(f))


class _QoaFile(C.Structure):
    _fields_ = [("channels", C.c_uint32), ("samplerate", C.c_uint32), ("samples", C.c_uint32), ("n_frames_pcm", C.c_uint64),
                ("n_qoa_frames", C.c_uint32), ("pcm", C.POINTER(C.c_int16))]


def qoa_decode_file(data):
    """Oracle QOA stream layer (oracle/qoa_lms.c): header + frame loop of QOADecoder.readSamples to the end of the stream."""
    buf = bytes(data)
    f = _QoaFile()
    fn = lib().afgo_qoa_decode_file
    fn.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(_QoaFile)]
    fn.restype = C.c_int
    rc = fn(buf, len(buf), C.byref(f))
    if rc != 0:
        return rc
    try:
        n = int(f.n_frames_pcm) * int(f.channels)
        pcm = np.ctypeslib.as_array(f.pcm, shape=(n,)).copy() if n else np.zeros(0, np.int16)
        return {"channels": int(f.channels), "samplerate": int(f.samplerate), "samples": int(f.samples), "pcm": pcm,
                "n_qoa_frames": int(f.n_qoa_frames)}
    finally:
        free = lib().afgo_qoa_file_free
        free.argtypes = [C.POINTER(_QoaFile)]
        free.restype = None
        free(C.byref(f))


# ------------------------------------------------------- CPU baseline pool ------
class BenchTask(C.Structure):
    _fields_ = [("codec", C.c_int32), ("n", C.c_uint32), ("channels", C.c_uint32), ("bs0", C.c_uint16), ("bs1", C.c_uint16),
                ("a", C.c_void_p), ("b", C.c_void_p), ("c", C.c_void_p), ("d", C.c_void_p), ("out_floats", C.c_uint64)]


def bench_task(codec, n, channels, bs0, bs1, a, b, c, d, out_floats):
    """One file for oracle/cpu_bench.c (the arrays must stay alive while the pool runs)."""
    def ptr(x):
        if x is None:
            return None
        assert x.flags.c_contiguous
        return x.ctypes.data
    return BenchTask(codec, n, channels, bs0, bs1, ptr(a), ptr(b), ptr(c), ptr(d), out_floats)


def bench_run(tasks, repeats, threads):
    """afgo_bench_run: (wall seconds, CPU seconds summed over the workers, output values produced)."""
    L = lib()
    L.afgo_bench_run.argtypes = [C.POINTER(BenchTask), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.afgo_bench_run.restype = C.c_double
    arr = (BenchTask * len(tasks))(*tasks)
    cpu, samples = C.c_double(0), C.c_uint64(0)
    wall = L.afgo_bench_run(arr, len(tasks), int(repeats), int(threads), C.byref(cpu), C.byref(samples))
    if wall < 0:
        raise RuntimeError("afgo_bench_run failed")
    return wall, cpu.value, samples.value


# ------------------------------------------------------------ WAV out ------
RAND_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)


def wav_pcm(x, bits, dither=None, rng_max=0x7fffffff):
    """The integer samples WAVEncoder.writeSamples writes for float input (wav.d:474-527).  dither: None = off,
    "libc" = the reference's rand(), or a callable returning draws in [0, rng_max]."""
    x = np.ascontiguousarray(x, np.float32).reshape(-1)
    out = np.zeros(x.size, np.int32)
    L = lib()
    L.afgo_wav_pcm.argtypes = [f32p, C.c_int, C.c_int, C.c_int, RAND_FN, C.c_void_p, C.c_double, i32p]
    cb = RAND_FN() if dither in (None, "libc") else RAND_FN(lambda _u: int(dither()))
    assert L.afgo_wav_pcm(x, x.size, bits, int(dither is not None), cb, None, float(rng_max), out) == 0
    return out
