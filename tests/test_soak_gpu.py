"""A short run of tools/soak_damaged.py: randomly damaged QOA / FLAC / Ogg Vorbis files through afg_batch_decode, against the
oracle's decode of the same damaged bytes (QOA / FLAC bit for bit, MP3 / Vorbis within tolerance, Opus within one int16 step), in both
numeric modes."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


def soak(seed, headers=False):
    import soak_damaged
    before = soak_damaged.HDR
    soak_damaged.HDR = headers
    try:
        ok, rejected, bad = soak_damaged.run(4, seed)
    finally:
        soak_damaged.HDR = before
    assert bad == 0 and ok >= 20


def test_damaged_files_exact_mode(gpu):
    soak(7)


@pytest.mark.numeric_tolerance
def test_damaged_files_default_mode(gpu):
    soak(8)


@pytest.mark.numeric_tolerance
def test_files_with_damaged_headers_too(gpu):
    """AFG_SOAK_HDR: the damage may land in the stream headers (the class that found the incomplete Vorbis code books)"""
    soak(9, headers=True)
