"""The C-ABI library loads and exports every symbol include/afg.h declares (no compute calls:
this runs without a GPU).  Also checks that the product never links or loads the oracle."""
import ctypes as C
import os
import re
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "afg.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(afg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import afgpu
    lib = afgpu.lib()
    names = declared_symbols()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(afgpu.ABI_SYMBOLS) == names          # the Python binding knows all of them
    assert lib.afg_abi_version() == 2
    assert lib.afg_status_string(0) == b"ok"
    assert lib.afg_status_string(-2) == b"no usable gfx950 device"


def test_d_binding_declares_the_headers_symbols():
    """bindings/d/afgpu.d (what a maintainer of the reference imports; never compiled here: no D compiler) declares exactly the
    functions include/afg.h declares."""
    text = open(os.path.join(ROOT, "bindings", "d", "afgpu.d")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    in_d = sorted(set(re.findall(r"\b(afg_[a-z0-9_]+)\s*\(", text)))
    names = declared_symbols()
    assert [n for n in names if n not in in_d] == [], "declared in afg.h, missing from the D binding"
    assert [n for n in in_d if n not in names] == [], "in the D binding, not in afg.h"


def test_dev_options_are_set_by_call_not_by_environment():
    """afg_dev_option (afg.h): the test hooks are named options of a call; the library holds no getenv for them (strings of
    the binary), and an unknown name is refused."""
    import afgpu
    lib = afgpu.lib()
    assert lib.afg_dev_option(b"celt_path", 3) == 0 and lib.afg_dev_option(b"celt_path", -1) == 0
    assert lib.afg_dev_option(b"no_such_option", 1) != 0
    so = open(os.path.join(ROOT, "audio-formats_amd", "lib", "libafg_hip.so"), "rb").read()
    for name in (b"AFG_CELT_DE_SEQ", b"AFG_CELT_DE_DUO", b"AFG_CELT_PATH", b"AFG_CELT_SEG_RECS", b"AFG_CELT_WHOLE_FRAMES", b"AFG_VORBIS_SINGLE",
                 b"AFG_MP3_CHUNKS", b"AFG_MP3_FLOAT_UPLOAD", b"AFG_VORBIS_HOST_FLOOR", b"AFG_FLAC_HOST_RES32"):
        assert name not in so, name


def test_record_layouts_match_the_header():
    import afgpu
    assert afgpu.FLAC_SUBFRAME_DTYPE.itemsize == 68 and afgpu.FLAC_FRAME_DTYPE.itemsize == 32
    assert afgpu.FLAC_FRAME_DTYPE.fields["sf_index"][1] == 20 and afgpu.FLAC_FRAME_DTYPE.fields["channels"][1] == 24
    assert int(afgpu.mp3_flags(3, 2, 1)) == 3 | (2 << 8) | (2 << 16)
    assert afgpu.FLAC_FRAME_DTYPE.fields["res16"][1] == 27
    assert afgpu.VORBIS_FLOOR_PACKET_DTYPE.itemsize == 32 and afgpu.VORBIS_FLOOR_CURVE_DTYPE.itemsize == 8
    assert afgpu.VORBIS_FLOOR_PACKET_DTYPE.fields["curve_index"][1] == 16 and afgpu.VORBIS_FLOOR_PACKET_DTYPE.fields["n_steps"][1] == 24


def test_fails_loudly_without_a_device():
    import torch
    import afgpu
    if torch.cuda.is_available():
        return
    try:
        afgpu.Mp3Plan([4], [2])
    except afgpu.AfgError as e:
        assert "no CPU fallback" in str(e) or "no usable gfx950 device" in str(e)
    else:
        raise AssertionError("plan creation must fail without a GPU: there is no CPU path")


def test_product_does_not_depend_on_the_oracle():
    lib = os.path.join(ROOT, "audio-formats_amd", "lib", "libafg_hip.so")
    needed = subprocess.check_output(["readelf", "-d", lib], text=True)
    assert "oracle" not in needed
    syms = subprocess.check_output(["nm", "-D", lib], text=True)
    assert "afgo_" not in syms
    for dirpath, _, files in os.walk(os.path.join(ROOT, "audio-formats_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oraclelib" not in text and "liboracle" not in text, f


def test_numeric_mode_switch_needs_no_device(monkeypatch):
    """afg_set_numeric_mode / afg_get_numeric_mode (afg.h): tolerance by default, AFG_NUMERIC decides until the call, the call
    wins afterwards, AFG_NUMERIC_FROM_ENV hands the choice back; an unknown mode is refused."""
    import afgpu
    lib = afgpu.lib()
    monkeypatch.delenv("AFG_NUMERIC", raising=False)
    assert afgpu.set_numeric_mode(afgpu.NUMERIC_FROM_ENV) in (afgpu.NUMERIC_EXACT, afgpu.NUMERIC_TOLERANCE)
    assert afgpu.get_numeric_mode() == afgpu.NUMERIC_TOLERANCE
    monkeypatch.setenv("AFG_NUMERIC", "exact")
    assert afgpu.get_numeric_mode() == afgpu.NUMERIC_EXACT
    assert afgpu.set_numeric_mode(afgpu.NUMERIC_TOLERANCE) == afgpu.NUMERIC_EXACT
    assert afgpu.get_numeric_mode() == afgpu.NUMERIC_TOLERANCE                  # the call outranks the environment
    assert lib.afg_set_numeric_mode(7) < 0 and b"unknown mode" in lib.afg_last_error()
    assert afgpu.get_numeric_mode() == afgpu.NUMERIC_TOLERANCE
    afgpu.set_numeric_mode(afgpu.NUMERIC_FROM_ENV)
    assert afgpu.get_numeric_mode() == afgpu.NUMERIC_EXACT
