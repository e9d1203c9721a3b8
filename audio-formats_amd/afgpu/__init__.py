"""afgpu -- Python host binding of the MI355X audio-decode transform path.

Thin ctypes view of ``audio-formats_amd/lib/libafg_hip.so`` (the C ABI declared in
``include/afg.h``).  PyTorch is used only as plumbing (device memory, streams,
``torch.distributed``); tensors cross the boundary as raw device pointers.

There is no CPU fallback: if the library or a gfx950 device is missing, every
entry point raises ``AfgError``.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("AFG_LIB_PATH", os.path.join(PKG_ROOT, "lib", "libafg_hip.so"))   # override: A/B builds only

MP3_STATE_FLOATS = 1536
VORBIS_LONG, VORBIS_PREV, VORBIS_NEXT = 1, 2, 4


def VORBIS_NZ_EIGHTHS(e):
    """afg.h AFG_VORBIS_NZ_EIGHTHS: a long packet's declaration that only its first e eighths may be nonzero."""
    return (int(e) + 1) << 4

FLAC_INDEPENDENT, FLAC_LEFT_SIDE, FLAC_RIGHT_SIDE, FLAC_MID_SIDE = 0, 8, 9, 10

FLAC_SUBFRAME_DTYPE = np.dtype([("coef", np.int16, (32,)), ("order", np.uint8), ("shift", np.uint8),
                                ("wasted", np.uint8), ("use64", np.uint8)], align=True)
VORBIS_FLOOR_PACKET_DTYPE = np.dtype([("spec_off", np.uint64), ("n2", np.uint32), ("channels", np.uint32), ("curve_index", np.uint32),
                                      ("step_off", np.uint32), ("n_steps", np.uint32), ("pad", np.uint32)])
VORBIS_FLOOR_CURVE_DTYPE = np.dtype([("point_off", np.uint32), ("n_points", np.uint32)])
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
QOA_ENC_STREAM_DTYPE = np.dtype([("pcm_off", np.uint64), ("out_off", np.uint64), ("samples", np.uint32),
                                 ("samplerate", np.uint32), ("channels", np.uint8), ("pad", np.uint8, 7)])
assert QOA_ENC_STREAM_DTYPE.itemsize == 32
WAV_S8, WAV_S16LE, WAV_S24LE, WAV_FP32LE, WAV_FP64LE = range(5)

# every symbol include/afg.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "afg_abi_version", "afg_status_string", "afg_last_error", "afg_device_count", "afg_device_name",
    "afg_set_numeric_mode", "afg_get_numeric_mode", "afg_dev_option",
    "afg_mp3_plan_create", "afg_mp3_plan_destroy", "afg_mp3_plan_blocks", "afg_mp3_plan_segments",
    "afg_mp3_transform_hip", "afg_mp3_requant_hip", "afg_mp3_parse_q", "afg_mp3_parsed_q_free", "afg_mp3_qtables",
    "afg_vorbis_plan_create", "afg_vorbis_plan_destroy", "afg_vorbis_plan_packets",
    "afg_vorbis_plan_spec_floats", "afg_vorbis_plan_out_floats", "afg_vorbis_plan_offsets",
    "afg_vorbis_transform_hip", "afg_vorbis_floor_hip", "afg_vorbis_parse_r", "afg_vorbis_parsed_r_free",
    "afg_flac_transform_hip", "afg_flac_variants", "afg_flac_transform_variants_hip",
    "afg_qoa_transform_hip",
    "afg_celt_transform_hip", "afg_celt_transform_streams_hip",
    "afg_open_from_memory", "afg_is_error", "afg_error_message", "afg_get_format", "afg_get_num_channels",
    "afg_get_length_in_frames", "afg_get_samplerate", "afg_read_samples_float", "afg_close",
    "afg_can_seek", "afg_seek_position", "afg_tell_position",
    "afg_flac_parse", "afg_flac_parsed_free", "afg_qoa_parse", "afg_mp3_parse", "afg_mp3_parsed_free", "afg_vorbis_parse", "afg_vorbis_parsed_free",
    "afg_opus_parse", "afg_opus_parsed_free", "afg_opus_output_gain_hip",
    "afg_batch_decode", "afg_batch_free", "afg_batch_decode_ex", "afg_set_device", "afg_get_device", "afg_host_pool_trim",
    "afg_device_malloc", "afg_device_free", "afg_memcpy_h2d", "afg_memcpy_d2h", "afg_stream_synchronize",
    "afg_copy_probe_hip", "afg_lds_fill_probe_hip",
    "afg_qoa_encoded_size", "afg_qoa_encode_hip", "afg_wav_encoded_size", "afg_wav_encode", "afg_wav_encode_dithered",
    "afg_opus_output_hip",
]


class AfgError(RuntimeError):
    pass


# AudioFileFormat (stream.d:36-47), in the reference's order
FORMAT_NAMES = ["wav", "mp3", "flac", "ogg", "opus", "qoa", "mod", "xm", "unknown"]
FORMAT_WAV, FORMAT_MP3, FORMAT_FLAC, FORMAT_OGG, FORMAT_OPUS, FORMAT_QOA, FORMAT_MOD, FORMAT_XM, FORMAT_UNKNOWN = range(9)
UNKNOWN_LENGTH = -1   # audiostreamUnknownLength, stream.d:90


class FlacParsed(C.Structure):
    _fields_ = [("sample_rate", C.c_uint32), ("channels", C.c_uint32), ("bps", C.c_uint32), ("max_block", C.c_uint32),
                ("total_samples", C.c_uint64), ("n_frames", C.c_uint64), ("n_subframes", C.c_uint64),
                ("n_res", C.c_uint64), ("out_samples", C.c_uint64), ("frames", C.c_void_p),
                ("subframes", C.c_void_p), ("res", C.c_void_p), ("owner", C.c_void_p)]


class Mp3Parsed(C.Structure):
    _fields_ = [("channels", C.c_int32), ("hz", C.c_int32), ("tagged", C.c_int32), ("start_delay", C.c_int32),
                ("detected_samples", C.c_uint64), ("declared_samples", C.c_uint64), ("pcm_samples", C.c_uint64),
                ("n_runs", C.c_uint64), ("n_blocks", C.c_uint64), ("n_copies", C.c_uint64),
                ("run_granules", C.c_void_p), ("coef", C.c_void_p), ("flags", C.c_void_p), ("copies", C.c_void_p),
                ("owner", C.c_void_p)]


class Mp3ParsedQ(C.Structure):
    _fields_ = [("base", Mp3Parsed), ("n_granules", C.c_uint64), ("n_sdesc", C.c_uint64), ("q", C.c_void_p),
                ("granules", C.c_void_p), ("sdesc", C.c_void_p)]


MP3_QGRANULE_DTYPE = np.dtype([("q_off", np.uint64), ("coef_off", np.uint64), ("sdesc", np.uint32), ("nch", np.uint8),
                               ("stereo", np.uint8), ("table", np.uint8, (2,)), ("scale", np.float32, (2, 40))], align=True)
MP3_SDESC_DTYPE = np.dtype([("type", np.uint8, (40,)), ("fl", np.float32, (40,)), ("fr", np.float32, (40,))], align=True)
assert MP3_QGRANULE_DTYPE.itemsize == 344 and MP3_SDESC_DTYPE.itemsize == 360


class VorbisParsed(C.Structure):
    _fields_ = [("channels", C.c_int32), ("blocksize0", C.c_int32), ("blocksize1", C.c_int32), ("sample_rate", C.c_uint32),
                ("total_samples", C.c_uint32), ("n_packets", C.c_uint64), ("spec_floats", C.c_uint64),
                ("pcm_frames", C.c_uint64), ("pflags", C.c_void_p), ("spec", C.c_void_p), ("take_from", C.c_void_p),
                ("take_count", C.c_void_p), ("owner", C.c_void_p)]


class VorbisParsedR(C.Structure):
    _fields_ = [("base", VorbisParsed), ("n_curves", C.c_uint64), ("n_points", C.c_uint64), ("n_steps", C.c_uint64),
                ("packets", C.c_void_p), ("curves", C.c_void_p), ("points", C.c_void_p), ("steps", C.c_void_p)]


class OpusParsed(C.Structure):
    _fields_ = [("channels", C.c_int32), ("preskip", C.c_int32), ("gain_i", C.c_int32), ("error", C.c_int32),
                ("gain", C.c_float), ("pad", C.c_int32), ("declared_frames", C.c_int64), ("pcm_frames", C.c_uint64),
                ("n_frames", C.c_uint64), ("n_coeffs", C.c_uint64), ("frames", C.c_void_p), ("coeffs", C.c_void_p),
                ("owner", C.c_void_p)]


class BatchItem(C.Structure):
    _fields_ = [("status", C.c_int), ("message", C.c_char_p), ("format", C.c_int), ("channels", C.c_int),
                ("samplerate", C.c_float), ("frames", C.c_int64), ("pcm", C.POINTER(C.c_float))]


class BatchOpts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_threads", C.c_int), ("n_devices", C.c_int), ("devices", C.POINTER(C.c_int))]


class BatchResult(C.Structure):
    _fields_ = [("n_files", C.c_int), ("items", C.POINTER(BatchItem)), ("owner", C.c_void_p)]


_lib = None
RAND_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)      # afg_rand_fn


def mp3_flags(block_type=0, n_long_bands=0, aa_bands=31):
    """AFG_MP3_FLAGS of include/afg.h."""
    return np.uint32(block_type | (n_long_bands << 8) | ((aa_bands + 1) << 16))


# The library's test hooks (afg.h: afg_dev_option) under the environment-variable names the test-suite has always used:
# THIS module reads the variables -- the C library reads none -- and hands changed values over before the next call.
_DEV_ENV = {"AFG_CELT_PATH": ("celt_path", {"stream": 1, "split": 2, "walk": 3}), "AFG_CELT_DE_SEQ": ("celt_de_seq", None),
            "AFG_CELT_DE_DUO": ("celt_de_duo", None), "AFG_CELT_SEG_RECS": ("celt_seg_recs", None),
            "AFG_CELT_WHOLE_FRAMES": ("celt_whole_frames", None), "AFG_VORBIS_SINGLE": ("vorbis_single", None),
            "AFG_MP3_CHUNKS": ("mp3_chunks", None), "AFG_MP3_FLOAT_UPLOAD": ("mp3_float_upload", None),
            "AFG_VORBIS_HOST_FLOOR": ("vorbis_host_floor", None), "AFG_FLAC_HOST_RES32": ("flac_host_res32", None),
            "AFG_VORBIS_SEG_PACKETS": ("vorbis_seg_packets", None), "AFG_BATCH_GROUPS": ("batch_groups", None)}
_dev_seen = {}


def _sync_dev_options(L):
    for env, (name, words) in _DEV_ENV.items():
        raw = os.environ.get(env)
        if _dev_seen.get(env, None) == raw and env in _dev_seen:
            continue
        _dev_seen[env] = raw
        if raw is None:
            value = -1
        elif words is not None:
            value = words.get(raw, -1)
        else:
            try:
                value = int(raw)
            except ValueError:
                value = 1
        L.afg_dev_option(name.encode(), value)


def lib():
    """Load the C-ABI library (once).  torch is imported first so that both share one HIP runtime."""
    global _lib
    if _lib is not None:
        _sync_dev_options(_lib)
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AfgError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback)")
    try:
        import torch  # noqa: F401  (loads libamdhip64 so the C ABI binds to the same runtime)
    except Exception:  # pragma: no cover - torch is plumbing only
        pass
    L = C.CDLL(LIB_PATH)
    vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
    L.afg_abi_version.restype = C.c_int
    L.afg_status_string.restype = C.c_char_p
    L.afg_status_string.argtypes = [C.c_int]
    L.afg_last_error.restype = C.c_char_p
    L.afg_device_count.restype = C.c_int
    L.afg_device_name.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
    L.afg_opus_output_hip.argtypes = [u64, vp, vp, vp, vp]
    L.afg_opus_output_gain_hip.argtypes = [u64, vp, C.c_float, vp, vp, vp]
    L.afg_opus_parse.argtypes = [vp, C.c_size_t, C.POINTER(OpusParsed)]
    L.afg_opus_parsed_free.argtypes = [C.POINTER(OpusParsed)]
    L.afg_opus_parsed_free.restype = None
    L.afg_qoa_encoded_size.argtypes = [u32, u32]
    L.afg_qoa_encoded_size.restype = u64
    L.afg_qoa_encode_hip.argtypes = [u32, vp, vp, vp, vp, vp]
    L.afg_wav_encoded_size.argtypes = [u64, u32, C.c_int]
    L.afg_wav_encoded_size.restype = u64
    L.afg_wav_encode.argtypes = [vp, u64, u32, u32, C.c_int, vp, u64]
    L.afg_wav_encode.restype = u64
    L.afg_wav_encode_dithered.argtypes = [vp, u64, u32, u32, C.c_int, RAND_FN, vp, u32, vp, u64]
    L.afg_wav_encode_dithered.restype = u64
    L.afg_mp3_plan_create.argtypes = [C.POINTER(vp), u32, vp, vp, u32]
    L.afg_mp3_plan_destroy.argtypes = [vp]
    L.afg_mp3_plan_destroy.restype = None
    L.afg_mp3_plan_blocks.argtypes = [vp]
    L.afg_mp3_plan_blocks.restype = u64
    L.afg_mp3_plan_segments.argtypes = [vp]
    L.afg_mp3_plan_segments.restype = u32
    L.afg_mp3_transform_hip.argtypes = [vp, vp, vp, vp, vp, vp]
    L.afg_vorbis_plan_create.argtypes = [C.POINTER(vp), u32, vp, vp, vp, vp, vp, u32]
    L.afg_vorbis_plan_destroy.argtypes = [vp]
    L.afg_vorbis_plan_destroy.restype = None
    for fn in (L.afg_vorbis_plan_packets, L.afg_vorbis_plan_spec_floats, L.afg_vorbis_plan_out_floats):
        fn.argtypes = [vp]
        fn.restype = u64
    L.afg_vorbis_plan_offsets.argtypes = [vp, vp, vp]
    L.afg_vorbis_transform_hip.argtypes = [vp, vp, vp, vp]
    L.afg_vorbis_floor_hip.argtypes = [u64, vp, vp, vp, vp, vp, vp]
    L.afg_flac_transform_hip.argtypes = [u64, vp, vp, vp, vp, vp, vp]
    L.afg_flac_variants.argtypes = [u64, vp, vp]
    L.afg_flac_variants.restype = u32
    L.afg_flac_transform_variants_hip.argtypes = [u64, vp, vp, vp, vp, vp, u32, vp]
    L.afg_qoa_transform_hip.argtypes = [u64, vp, vp, vp, vp, vp]
    L.afg_celt_transform_hip.argtypes = [u32, vp, vp, vp, vp, vp, vp]
    L.afg_celt_transform_streams_hip.argtypes = [u32, vp, vp, vp, vp, vp, vp, vp]
    L.afg_open_from_memory.argtypes = [vp, C.c_size_t]
    L.afg_open_from_memory.restype = vp
    L.afg_is_error.argtypes = [vp]
    L.afg_error_message.argtypes = [vp]
    L.afg_error_message.restype = C.c_char_p
    L.afg_get_format.argtypes = [vp]
    L.afg_get_num_channels.argtypes = [vp]
    L.afg_get_length_in_frames.argtypes = [vp]
    L.afg_get_length_in_frames.restype = C.c_int64
    L.afg_get_samplerate.argtypes = [vp]
    L.afg_get_samplerate.restype = C.c_float
    L.afg_read_samples_float.argtypes = [vp, vp, C.c_int]
    L.afg_close.argtypes = [vp]
    L.afg_can_seek.argtypes = [vp]
    L.afg_seek_position.argtypes = [vp, C.c_int]
    L.afg_tell_position.argtypes = [vp]
    L.afg_close.restype = None
    L.afg_flac_parse.argtypes = [vp, C.c_size_t, C.POINTER(FlacParsed)]
    L.afg_flac_parsed_free.argtypes = [C.POINTER(FlacParsed)]
    L.afg_flac_parsed_free.restype = None
    L.afg_mp3_parse.argtypes = [vp, C.c_size_t, C.POINTER(Mp3Parsed)]
    L.afg_mp3_parsed_free.argtypes = [C.POINTER(Mp3Parsed)]
    L.afg_mp3_parsed_free.restype = None
    L.afg_mp3_parse_q.argtypes = [vp, C.c_size_t, C.POINTER(Mp3ParsedQ)]
    L.afg_mp3_parsed_q_free.argtypes = [C.POINTER(Mp3ParsedQ)]
    L.afg_mp3_parsed_q_free.restype = None
    L.afg_mp3_requant_hip.argtypes = [u64, vp, vp, vp, vp, vp]
    L.afg_vorbis_parse.argtypes = [vp, C.c_size_t, C.POINTER(VorbisParsed)]
    L.afg_vorbis_parsed_free.argtypes = [C.POINTER(VorbisParsed)]
    L.afg_vorbis_parsed_free.restype = None
    L.afg_vorbis_parse_r.argtypes = [vp, C.c_size_t, C.POINTER(VorbisParsedR)]
    L.afg_vorbis_parsed_r_free.argtypes = [C.POINTER(VorbisParsedR)]
    L.afg_vorbis_parsed_r_free.restype = None
    L.afg_qoa_parse.argtypes = [vp, C.c_size_t, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), vp, C.c_size_t,
                                C.POINTER(C.c_size_t)]
    L.afg_batch_decode.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(BatchResult)]
    L.afg_batch_free.argtypes = [C.POINTER(BatchResult)]
    L.afg_batch_decode_ex.argtypes = [vp, vp, C.c_int, C.POINTER(BatchOpts), C.POINTER(BatchResult)]
    L.afg_set_device.argtypes = [C.c_int]
    L.afg_get_device.restype = C.c_int
    L.afg_host_pool_trim.restype = u64
    L.afg_batch_free.restype = None
    L.afg_device_malloc.argtypes = [C.POINTER(vp), C.c_size_t]
    L.afg_device_free.argtypes = [vp]
    L.afg_memcpy_h2d.argtypes = [vp, vp, C.c_size_t, vp]
    L.afg_memcpy_d2h.argtypes = [vp, vp, C.c_size_t, vp]
    L.afg_stream_synchronize.argtypes = [vp]
    L.afg_copy_probe_hip.argtypes = [vp, vp, C.c_size_t, vp]
    L.afg_lds_fill_probe_hip.argtypes = [C.c_uint32, vp]
    L.afg_dev_option.argtypes = [C.c_char_p, C.c_int]
    _lib = L
    _sync_dev_options(L)
    return L


def check(rc):
    if rc != 0:
        L = lib()
        raise AfgError(f"afg: {L.afg_status_string(rc).decode()} ({rc}): {L.afg_last_error().decode()}")


def _ptr(t):
    """Device pointer of a torch tensor / int / None."""
    if t is None:
        return None
    if isinstance(t, int):
        return t
    assert t.is_cuda and t.is_contiguous(), "device tensors must be contiguous CUDA(HIP) tensors"
    return t.data_ptr()


def _stream(stream):
    if stream is None:
        import torch
        return torch.cuda.current_stream().cuda_stream
    return getattr(stream, "cuda_stream", stream)


def _np(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class Mp3Plan:
    """Batch description of the MP3 transform stage (afg_mp3_plan)."""

    def __init__(self, granules, channels, seg_granules=0):
        self.granules = _np(granules, np.uint32)
        self.channels = _np(channels, np.uint8)
        assert self.granules.shape == self.channels.shape
        self._h = C.c_void_p()
        check(lib().afg_mp3_plan_create(C.byref(self._h), len(self.granules), self.granules.ctypes.data,
                                        self.channels.ctypes.data, seg_granules))
        self.blocks = int(lib().afg_mp3_plan_blocks(self._h))
        self.segments = int(lib().afg_mp3_plan_segments(self._h))

    def transform(self, d_coef, d_flags, d_pcm, d_state=None, stream=None):
        """Enqueue the transform (no synchronisation).  Arguments are CUDA(HIP) tensors."""
        check(lib().afg_mp3_transform_hip(self._h, _ptr(d_coef), _ptr(d_flags), _ptr(d_pcm),
                                          _ptr(d_state), _stream(stream)))

    def close(self):
        if self._h:
            lib().afg_mp3_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VorbisPlan:
    """Batch description of the Vorbis transform stage (afg_vorbis_plan)."""

    def __init__(self, packets, channels, blocksize0, blocksize1, pflags, seg_packets=0):
        self.packets = _np(packets, np.uint32)
        self.channels = _np(channels, np.uint8)
        self.bs0 = _np(blocksize0, np.uint16)
        self.bs1 = _np(blocksize1, np.uint16)
        self.pflags = _np(pflags, np.uint8)
        assert int(self.packets.sum()) == self.pflags.size
        self._h = C.c_void_p()
        check(lib().afg_vorbis_plan_create(C.byref(self._h), len(self.packets), self.packets.ctypes.data,
                                           self.channels.ctypes.data, self.bs0.ctypes.data,
                                           self.bs1.ctypes.data, self.pflags.ctypes.data, seg_packets))
        self.total_packets = int(lib().afg_vorbis_plan_packets(self._h))
        self.spec_floats = int(lib().afg_vorbis_plan_spec_floats(self._h))
        self.out_floats = int(lib().afg_vorbis_plan_out_floats(self._h))

    def offsets(self):
        so = np.zeros(self.total_packets, np.uint64)
        oo = np.zeros(self.total_packets, np.uint64)
        check(lib().afg_vorbis_plan_offsets(self._h, so.ctypes.data, oo.ctypes.data))
        return so, oo

    def transform(self, d_spec, d_out, stream=None):
        check(lib().afg_vorbis_transform_hip(self._h, _ptr(d_spec), _ptr(d_out), _stream(stream)))

    def close(self):
        if self._h:
            lib().afg_vorbis_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def flac_transform(n_frames, d_frames, d_subframes, d_res, d_out_i32=None, d_out_f32=None, stream=None, variants=None):
    """Enqueue the FLAC restore (afg_flac_transform_hip).  Records are uint8 CUDA tensors holding
    FLAC_FRAME_DTYPE / FLAC_SUBFRAME_DTYPE arrays.  `variants`: the mask flac_variants() computed from the host records
    (afg_flac_transform_variants_hip: only the populated instantiations are launched, side by side)."""
    if variants is None:
        check(lib().afg_flac_transform_hip(int(n_frames), _ptr(d_frames), _ptr(d_subframes), _ptr(d_res),
                                           _ptr(d_out_i32), _ptr(d_out_f32), _stream(stream)))
    else:
        check(lib().afg_flac_transform_variants_hip(int(n_frames), _ptr(d_frames), _ptr(d_subframes), _ptr(d_res),
                                                    _ptr(d_out_i32), _ptr(d_out_f32), int(variants), _stream(stream)))


def flac_variants(frames, subframes):
    """afg_flac_variants on host record arrays (FLAC_FRAME_DTYPE / FLAC_SUBFRAME_DTYPE): bit mask of the kernel
    instantiations the batch populates."""
    frames = np.ascontiguousarray(frames)
    subframes = np.ascontiguousarray(subframes)
    assert frames.dtype.itemsize == 32 and subframes.dtype.itemsize == 68
    return int(lib().afg_flac_variants(len(frames), frames.ctypes.data, subframes.ctypes.data))


def qoa_frames(file_bytes, out_base=0, byte_base=0):
    """Locate the frames of one QOA file (host side of qoa.d:413-486: magic, frame headers).
    Returns (QOA_FRAME_DTYPE array, channels, samplerate, total samples per channel)."""
    b = np.frombuffer(file_bytes, np.uint8)
    if b.size < 16 or bytes(b[:4]) != b"qoaf":
        raise AfgError("not a QOA file")
    total = int.from_bytes(bytes(b[4:8]), "big")
    recs, pos, out = [], 8, out_base
    first = int.from_bytes(bytes(b[8:16]), "big")
    channels, rate = (first >> 56) & 0xff, (first >> 32) & 0xffffff
    if not total or not channels or not rate or channels > 8:
        raise AfgError("not a QOA file")
    # the reader of qoa.d:455-534: a frame is header + LMS state + ceil(samples / 20) slices per channel from the cursor;
    # the frame-size field is checked (:477, :481-486), never used to find the next frame
    while b.size - pos >= 8 + 16 * channels:
        hdr = int.from_bytes(bytes(b[pos:pos + 8]), "big")
        ch, sr, smp, fsz = (hdr >> 56) & 0xff, (hdr >> 32) & 0xffffff, (hdr >> 16) & 0xffff, hdr & 0xffff
        num_slices = int((fsz - 8 - 16 * ch) / 8)                    # truncating division, as D's
        if b.size - pos - 8 < fsz - 8:
            break
        if ch != channels or sr != rate or smp * ch > num_slices * 20 or smp == 0 or smp > 5120:
            break
        used = 8 + 16 * ch + 8 * ch * ((smp + 19) // 20)
        if used > b.size - pos:
            break
        recs.append((byte_base + pos, out, smp, ch, [0] * 5))
        out += smp * ch
        pos += used
    return np.array(recs, QOA_FRAME_DTYPE), channels, rate, total


def qoa_transform(n_frames, d_frames, d_bytes, d_out_i16=None, d_out_f32=None, stream=None):
    """Enqueue the QOA frame decode (afg_qoa_transform_hip)."""
    check(lib().afg_qoa_transform_hip(int(n_frames), _ptr(d_frames), _ptr(d_bytes), _ptr(d_out_i16),
                                      _ptr(d_out_f32), _stream(stream)))


def opus_output(n_samples, d_in, d_out_i16=None, d_out_f32=None, stream=None, gain=None):
    """Enqueue OpusFile.readFrame's float -> int16 (-> float / 32767) conversion (afg_opus_output_hip); gain: the decoder's
    output gain is applied first (afg_opus_output_gain_hip)."""
    if gain is None:
        check(lib().afg_opus_output_hip(int(n_samples), _ptr(d_in), _ptr(d_out_i16), _ptr(d_out_f32), _stream(stream)))
    else:
        check(lib().afg_opus_output_gain_hip(int(n_samples), _ptr(d_in), float(gain), _ptr(d_out_i16), _ptr(d_out_f32), _stream(stream)))


def qoa_encoded_size(samples, channels):
    return int(lib().afg_qoa_encoded_size(int(samples), int(channels)))


def qoa_encode(n_streams, d_streams, d_out, d_pcm_i16=None, d_pcm_f32=None, stream=None):
    """Enqueue the QOA encoder (afg_qoa_encode_hip): d_streams is a device array of QOA_ENC_STREAM_DTYPE."""
    check(lib().afg_qoa_encode_hip(int(n_streams), _ptr(d_streams), _ptr(d_pcm_i16), _ptr(d_pcm_f32), _ptr(d_out),
                                   _stream(stream)))


def qoa_encode_layout(shapes, samplerate=44100):
    """Stream table for interleaved PCM blocks laid back to back: shapes = [(frames, channels), ...].
    Returns (QOA_ENC_STREAM_DTYPE array, total input samples, total output bytes)."""
    recs = np.zeros(len(shapes), QOA_ENC_STREAM_DTYPE)
    pcm = out = 0
    for i, (n, ch) in enumerate(shapes):
        recs[i] = (pcm, out, n, samplerate, ch, 0)
        pcm += n * ch
        out += (qoa_encoded_size(n, ch) + 7) & ~7
    return recs, pcm, out


def wav_encode(samples, samplerate, fmt=WAV_FP32LE, dither=None, rng_max=0x7fffffff):
    """Host WAV writer: samples float32 [frames, channels] -> file bytes.  dither=None: afg_wav_encode (no dither);
    dither="libc": TPDF dither from libc rand() as the reference; dither=callable: draws in [0, rng_max] from it."""
    x = np.ascontiguousarray(samples, np.float32)
    if x.ndim == 1:
        x = x[:, None]
    frames, ch = x.shape
    size = int(lib().afg_wav_encoded_size(frames, ch, int(fmt)))
    if not size:
        raise AfgError("afg_wav_encode: bad arguments")
    out = np.zeros(size, np.uint8)
    if dither is None:
        n = int(lib().afg_wav_encode(x.ctypes.data, frames, ch, int(samplerate), int(fmt), out.ctypes.data, size))
    else:
        cb = RAND_FN() if dither == "libc" else RAND_FN(lambda _user: int(dither()))
        n = int(lib().afg_wav_encode_dithered(x.ctypes.data, frames, ch, int(samplerate), int(fmt), cb, None, int(rng_max),
                                              out.ctypes.data, size))
    if n != size:
        raise AfgError("afg_wav_encode failed")
    return out.tobytes()


def celt_transform(n_chan, d_rec_base, d_recs, d_coeffs, d_out, d_states=None, stream=None, tail_stream=None):
    """Enqueue the CELT transform stage (afg_celt_transform_hip; with tail_stream afg_celt_transform_streams_hip: the
    per-sequence passes go to that stream behind an event, the caller joins)."""
    if tail_stream is None:
        check(lib().afg_celt_transform_hip(int(n_chan), _ptr(d_rec_base), _ptr(d_recs), _ptr(d_coeffs), _ptr(d_out),
                                           _ptr(d_states), _stream(stream)))
    else:
        check(lib().afg_celt_transform_streams_hip(int(n_chan), _ptr(d_rec_base), _ptr(d_recs), _ptr(d_coeffs), _ptr(d_out),
                                                   _ptr(d_states), _stream(stream), _stream(tail_stream)))


def flac_parse(file_bytes):
    """Host front-end only (afg_flac_parse): returns (info dict, frames, subframes, residual planes) as numpy
    copies of the transform-stage records.  Needs no device."""
    buf = bytes(file_bytes)
    out = FlacParsed()
    check(lib().afg_flac_parse(buf, len(buf), C.byref(out)))
    try:
        def view(ptr, count, dtype):
            if not count:
                return np.zeros(0, dtype)
            raw = (C.c_uint8 * (count * np.dtype(dtype).itemsize)).from_address(ptr)
            return np.frombuffer(raw, dtype=dtype, count=count).copy()
        info = {k: int(getattr(out, k)) for k in ("sample_rate", "channels", "bps", "max_block", "total_samples",
                                                  "out_samples")}
        return (info, view(out.frames, out.n_frames, FLAC_FRAME_DTYPE),
                view(out.subframes, out.n_subframes, FLAC_SUBFRAME_DTYPE), view(out.res, out.n_res, np.int32))
    finally:
        lib().afg_flac_parsed_free(C.byref(out))


def mp3_parse(file_bytes):
    """Host front-end only (afg_mp3_parse): (info dict, run_granules, coef [blocks, 576], flags, copies [n, 2])."""
    buf = bytes(file_bytes)
    out = Mp3Parsed()
    check(lib().afg_mp3_parse(buf, len(buf), C.byref(out)))
    try:
        def view(ptr, count, dtype):
            if not count:
                return np.zeros(0, dtype)
            raw = (C.c_uint8 * (count * np.dtype(dtype).itemsize)).from_address(ptr)
            return np.frombuffer(raw, dtype=dtype, count=count).copy()
        info = {k: int(getattr(out, k)) for k in ("channels", "hz", "tagged", "start_delay", "detected_samples",
                                                  "declared_samples", "pcm_samples")}
        return (info, view(out.run_granules, out.n_runs, np.uint32),
                view(out.coef, out.n_blocks * 576, np.float32).reshape(-1, 576), view(out.flags, out.n_blocks, np.uint32),
                view(out.copies, out.n_copies * 2, np.uint64).reshape(-1, 2))
    finally:
        lib().afg_mp3_parsed_free(C.byref(out))


def mp3_parse_q(file_bytes):
    """Host front-end in quantised mode (afg_mp3_parse_q): (info, run_granules, q int16 [blocks, 576], flags, copies,
    granule records, stereo descriptors).  Raises AfgError (unsupported) for streams the device requantiser does not cover."""
    buf = bytes(file_bytes)
    out = Mp3ParsedQ()
    check(lib().afg_mp3_parse_q(buf, len(buf), C.byref(out)))
    try:
        def view(ptr, count, dtype):
            if not count:
                return np.zeros(0, dtype)
            raw = (C.c_uint8 * (count * np.dtype(dtype).itemsize)).from_address(ptr)
            return np.frombuffer(raw, dtype=dtype, count=count).copy()
        b = out.base
        info = {k: int(getattr(b, k)) for k in ("channels", "hz", "tagged", "start_delay", "detected_samples",
                                                "declared_samples", "pcm_samples")}
        return (info, view(b.run_granules, b.n_runs, np.uint32), view(out.q, b.n_blocks * 576, np.int16).reshape(-1, 576),
                view(b.flags, b.n_blocks, np.uint32), view(b.copies, b.n_copies * 2, np.uint64).reshape(-1, 2),
                view(out.granules, out.n_granules, MP3_QGRANULE_DTYPE), view(out.sdesc, out.n_sdesc, MP3_SDESC_DTYPE))
    finally:
        lib().afg_mp3_parsed_q_free(C.byref(out))


def mp3_qtables():
    """afg_mp3_qtables: (band_of_line uint8 [24, 576], dst_of_src uint16 [24, 576], pow43 float32 [145])."""
    bol, dst, p43 = np.zeros((24, 576), np.uint8), np.zeros((24, 576), np.uint16), np.zeros(145, np.float32)
    fn = lib().afg_mp3_qtables
    fn.argtypes = [C.c_void_p] * 3
    fn.restype = None
    fn(bol.ctypes.data, dst.ctypes.data, p43.ctypes.data)
    return bol, dst, p43


def mp3_requant(n_granules, d_granules, d_q, d_sdesc, d_coef, stream=None):
    """Enqueue the MP3 requantisation (afg_mp3_requant_hip)."""
    check(lib().afg_mp3_requant_hip(int(n_granules), _ptr(d_granules), _ptr(d_q), _ptr(d_sdesc), _ptr(d_coef), _stream(stream)))


def vorbis_parse(file_bytes):
    """Host front-end only (afg_vorbis_parse): dict with channels, sample_rate, blocksize0/1, total_samples, pflags,
    spec, take_from, take_count, pcm_frames (numpy copies)."""
    buf = bytes(file_bytes)
    out = VorbisParsed()
    check(lib().afg_vorbis_parse(buf, len(buf), C.byref(out)))
    try:
        def view(ptr, count, dtype):
            if not count:
                return np.zeros(0, dtype)
            raw = (C.c_uint8 * (count * np.dtype(dtype).itemsize)).from_address(ptr)
            return np.frombuffer(raw, dtype=dtype, count=count).copy()
        n = int(out.n_packets)
        return {"channels": out.channels, "sample_rate": out.sample_rate, "blocksize0": out.blocksize0,
                "blocksize1": out.blocksize1, "total_samples": int(out.total_samples), "pcm_frames": int(out.pcm_frames),
                "pflags": view(out.pflags, n, np.uint8), "spec": view(out.spec, int(out.spec_floats), np.float32),
                "take_from": view(out.take_from, n, np.int32), "take_count": view(out.take_count, n, np.int32)}
    finally:
        lib().afg_vorbis_parsed_free(C.byref(out))


def vorbis_parse_r(file_bytes):
    """afg_vorbis_parse_r: vorbis_parse's dict with `spec` holding residue vectors, plus the inputs of vorbis_floor:
    fl_packets (VORBIS_FLOOR_PACKET_DTYPE), fl_curves (VORBIS_FLOOR_CURVE_DTYPE), fl_points int32 [n, 2], fl_steps uint8 [n, 2]."""
    buf = bytes(file_bytes)
    out = VorbisParsedR()
    check(lib().afg_vorbis_parse_r(buf, len(buf), C.byref(out)))
    try:
        def view(ptr, count, dtype):
            if not count:
                return np.zeros(0, dtype)
            raw = (C.c_uint8 * (count * np.dtype(dtype).itemsize)).from_address(ptr)
            return np.frombuffer(raw, dtype=dtype, count=count).copy()
        b = out.base
        n = int(b.n_packets)
        return {"channels": b.channels, "sample_rate": b.sample_rate, "blocksize0": b.blocksize0,
                "blocksize1": b.blocksize1, "total_samples": int(b.total_samples), "pcm_frames": int(b.pcm_frames),
                "pflags": view(b.pflags, n, np.uint8), "spec": view(b.spec, int(b.spec_floats), np.float32),
                "take_from": view(b.take_from, n, np.int32), "take_count": view(b.take_count, n, np.int32),
                "fl_packets": view(out.packets, n, VORBIS_FLOOR_PACKET_DTYPE),
                "fl_curves": view(out.curves, int(out.n_curves), VORBIS_FLOOR_CURVE_DTYPE),
                "fl_points": view(out.points, 2 * int(out.n_points), np.int32).reshape(-1, 2),
                "fl_steps": view(out.steps, 2 * int(out.n_steps), np.uint8).reshape(-1, 2)}
    finally:
        lib().afg_vorbis_parsed_r_free(C.byref(out))


def vorbis_floor(n_packets, d_packets, d_curves, d_points, d_steps, d_spec, stream=None):
    """afg_vorbis_floor_hip: inverse coupling + floor curves in place on the residue plane (device tensors; records as uint8
    views of the dtypes above)."""
    check(lib().afg_vorbis_floor_hip(int(n_packets), _ptr(d_packets), _ptr(d_curves), _ptr(d_points), _ptr(d_steps), _ptr(d_spec), _stream(stream)))


def opus_parse(file_bytes):
    """Host front-end only (afg_opus_parse): dict with channels, preskip, gain_i, gain, error, declared_frames, pcm_frames,
    frames (CELT_FRAME_DTYPE, channel 0's record per frame) and coeffs (numpy copies).  Raises AfgError for a stream that
    is not Ogg Opus or that holds SILK / hybrid packets."""
    buf = bytes(file_bytes)
    out = OpusParsed()
    check(lib().afg_opus_parse(buf, len(buf), C.byref(out)))
    try:
        def view(ptr, count, dtype):
            if not count:
                return np.zeros(0, dtype)
            raw = (C.c_uint8 * (count * np.dtype(dtype).itemsize)).from_address(ptr)
            return np.frombuffer(raw, dtype=dtype, count=count).copy()
        return {"channels": out.channels, "preskip": out.preskip, "gain_i": out.gain_i, "gain": float(np.float32(out.gain)),
                "error": bool(out.error), "declared_frames": int(out.declared_frames), "pcm_frames": int(out.pcm_frames),
                "frames": view(out.frames, int(out.n_frames), CELT_FRAME_DTYPE),
                "coeffs": view(out.coeffs, int(out.n_coeffs), np.float32)}
    finally:
        lib().afg_opus_parsed_free(C.byref(out))


def qoa_parse(file_bytes):
    """afg_qoa_parse: (QOA_FRAME_DTYPE array, channels, samplerate, samples per channel)."""
    buf = bytes(file_bytes)
    ch, sr, smp, n = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_size_t()
    check(lib().afg_qoa_parse(buf, len(buf), C.byref(ch), C.byref(sr), C.byref(smp), None, 0, C.byref(n)))
    frames = np.zeros(n.value, QOA_FRAME_DTYPE)
    check(lib().afg_qoa_parse(buf, len(buf), None, None, None, frames.ctypes.data, n.value, None))
    return frames, ch.value, sr.value, smp.value


class AudioStream:
    """The reading half of the reference's AudioStream (stream.d:102-637) over afg_open_from_memory: same method
    names, same never-throw / error-state contract (stream.d:31-33)."""

    def __init__(self):
        self._h = None
        self._keep = None

    def openFromMemory(self, data):
        self.cleanUp()
        self._keep = bytes(data)
        self._h = lib().afg_open_from_memory(self._keep, len(self._keep))

    def cleanUp(self):
        if self._h:
            lib().afg_close(self._h)
        self._h = None

    __del__ = cleanUp

    def isError(self):
        return bool(lib().afg_is_error(self._h))

    def errorMessage(self):
        m = lib().afg_error_message(self._h)
        return None if m is None else m.decode()

    def getFormat(self):
        return int(lib().afg_get_format(self._h))

    def getNumChannels(self):
        return int(lib().afg_get_num_channels(self._h))

    def getLengthInFrames(self):
        return int(lib().afg_get_length_in_frames(self._h))

    def getSamplerate(self):
        return float(lib().afg_get_samplerate(self._h))

    def canSeek(self):
        return bool(lib().afg_can_seek(self._h))

    def seekPosition(self, frame):
        return bool(lib().afg_seek_position(self._h, int(frame)))

    def tellPosition(self):
        return int(lib().afg_tell_position(self._h))

    def readSamplesFloat(self, out):
        """out: float32 numpy array whose size is a multiple of the channel count; returns frames read."""
        ch = max(1, self.getNumChannels())
        assert out.dtype == np.float32 and out.flags.c_contiguous and out.size % ch == 0
        return int(lib().afg_read_samples_float(self._h, out.ctypes.data, out.size // ch))


class BatchDecoded:
    """Result of afg_batch_decode kept in the library's (page-locked) result plane: ``items[i]`` are dicts whose
    ``pcm`` arrays are views, valid until ``close()`` (or the end of a ``with`` block)."""

    def __init__(self, files, n_threads=0, devices=None):
        """devices: None = the current device; "all" = every visible device; or a list of device indices."""
        self.devices = devices
        self._bufs = [bytes(f) for f in files]
        n = len(self._bufs)
        self._ptrs = (C.c_char_p * max(n, 1))(*self._bufs)
        self._lens = (C.c_size_t * max(n, 1))(*[len(b) for b in self._bufs])
        self._res = BatchResult()
        self._open = False
        self.n_threads = n_threads
        self._items = None

    def run(self):
        """The timed part: host parse + device restore + copy back -- the C call and nothing else (the per-file views of
        `items` are made when they are first asked for: two thousand numpy views cost more than some of these calls)."""
        self.close()
        opts = BatchOpts(C.sizeof(BatchOpts), self.n_threads, 0, None)
        if self.devices == "all":
            opts.n_devices = -1
        elif self.devices is not None:
            devs = (C.c_int * len(self.devices))(*[int(d) for d in self.devices])
            opts.n_devices, opts.devices = len(self.devices), devs
        check(lib().afg_batch_decode_ex(self._ptrs, self._lens, len(self._bufs), C.byref(opts), C.byref(self._res)))
        self._open = True
        self._items = None
        return self

    @property
    def items(self):
        if self._items is None:
            self._items = []
            if self._open:
                for i in range(self._res.n_files):
                    it = self._res.items[i]
                    cnt = it.frames * it.channels
                    pcm = (np.ctypeslib.as_array(it.pcm, shape=(cnt,)).reshape(-1, max(1, it.channels))
                           if cnt and it.pcm else None)
                    self._items.append({"status": it.status, "message": None if it.message is None else it.message.decode(),
                                        "format": it.format, "channels": it.channels, "samplerate": it.samplerate,
                                        "frames": it.frames, "pcm": pcm})
        return self._items

    def close(self):
        if self._open:
            lib().afg_batch_free(C.byref(self._res))
            self._open = False
        self._items = None

    def __enter__(self):
        return self.run() if not self._open else self

    def __exit__(self, *exc):
        self.close()

    __del__ = close


def set_device(device):
    """afg_set_device: make `device` current for the calling host thread (and keep torch's notion in step)."""
    check(lib().afg_set_device(int(device)))
    try:
        import torch
        torch.cuda.set_device(int(device))
    except Exception:  # pragma: no cover - torch is plumbing only
        pass


def get_device():
    d = int(lib().afg_get_device())
    if d < 0:
        check(d)
    return d


def batch_decode(files, n_threads=0, devices=None):
    """afg_batch_decode(_ex): list of dicts (status, message, format, channels, samplerate, frames, pcm ndarray copy)."""
    with BatchDecoded(files, n_threads, devices) as res:
        return [dict(it, pcm=None if it["pcm"] is None else it["pcm"].copy()) for it in res.items]


def copy_probe(d_dst, d_src, nbytes, stream=None):
    """Enqueue the streaming copy used to measure the practical HBM copy ceiling (afg_copy_probe_hip)."""
    check(lib().afg_copy_probe_hip(_ptr(d_dst), _ptr(d_src), int(nbytes), _stream(stream)))


def lds_fill(word=0x7fc00000, stream=None):
    """Test aid (afg_lds_fill_probe_hip): leave `word` (default: NaN) in the LDS of every compute unit."""
    check(lib().afg_lds_fill_probe_hip(int(word), _stream(stream)))


NUMERIC_FROM_ENV, NUMERIC_EXACT, NUMERIC_TOLERANCE = -1, 0, 1


def set_numeric_mode(mode):
    """afg_set_numeric_mode: NUMERIC_EXACT (the reference's expression trees, bit for bit) or NUMERIC_TOLERANCE (default:
    within 1e-5 RMS; the Opus/CELT stage may re-associate); NUMERIC_FROM_ENV hands the choice back to AFG_NUMERIC.
    Returns the mode that was in effect before."""
    prev = lib().afg_set_numeric_mode(int(mode))
    if prev < 0:
        check(prev)
    return prev


def get_numeric_mode():
    return int(lib().afg_get_numeric_mode())


def device_count():
    return int(lib().afg_device_count())


def device_name(device=0):
    buf = C.create_string_buffer(256)
    check(lib().afg_device_name(device, buf, 256))
    return buf.value.decode()
