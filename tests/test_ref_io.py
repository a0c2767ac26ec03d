"""The repository's .hdr reader and PNG16 writer against the reference's own (its vendored stb_image and lodepng,
compiled from /root/reference/include into oracle/_ref/libref_io.so), and against the fixture frozen from them.

The frozen fixture (tests/golden/hdr_decode.npz, made by tools/make_golden_hdr.py) is checked everywhere; the live
comparisons run only where the reference's sources exist."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as O
import ref_io_lib as R
import scenes as S

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from make_golden_hdr import write_flat_hdr  # noqa: E402

needs_ref = pytest.mark.skipif(not R.available(), reason="oracle/_ref/libref_io.so is built only where /root/reference exists")
GOLDEN = os.path.join(S.GOLDEN, "hdr_decode.npz")


def test_rgbe_decode_matches_frozen_reference_output(pkg, tmp_path):
    """Every exponent byte, several mantissas: the oracle's pixel decode and the product's file reader give the floats the
    reference's stbi_loadf gave (frozen), bit for bit; alpha is 1."""
    g = np.load(GOLDEN)
    rgbe, width, want = g["rgbe"], int(g["width"]), g["rgba"]
    got = np.zeros((rgbe.shape[0], 4), dtype=np.float32)
    O.lib().orc_rgbe_to_float(O.ptr(np.ascontiguousarray(rgbe)), rgbe.shape[0], O.ptr(got))
    assert np.array_equal(got.view(np.uint32), want.reshape(-1, 4).view(np.uint32))
    path = str(tmp_path / "flat.hdr")
    write_flat_hdr(path, rgbe, width)
    mine = pkg.hdr_load(path)
    assert mine.shape == want.shape
    assert np.array_equal(mine.view(np.uint32), want.view(np.uint32))
    assert (mine[..., 3] == 1.0).all()


@needs_ref
@pytest.mark.parametrize("kind", ["studio.hdr", "sunset.hdr", "forest.hdr"])
def test_hdr_reader_matches_reference_reader(pkg, kind):
    """The 2048x1024 run-length encoded maps the benchmark scenes use: this repository's reader and the reference's give
    the same floats."""
    path = S.ensure_asset(pkg, kind)
    mine, ref = pkg.hdr_load(path), R.hdr_load(path)
    assert mine.shape == ref.shape == (1024, 2048, 4)
    assert np.array_equal(mine.view(np.uint32), ref.view(np.uint32))


@needs_ref
def test_hdr_writer_round_trips_through_reference_reader(pkg, tmp_path):
    """A file this repository writes (run-length encoded scanlines) is read by the reference's reader to what this
    repository's reader returns, runs and literals included."""
    rng = np.random.default_rng(4)
    img = rng.random((17, 40, 3)).astype(np.float32) * np.float32(30.0)
    img[3:9, 5:30] = np.float32(2.5)              # long runs
    img[12] = 0.0                                  # a black scanline (exponent byte 0)
    path = str(tmp_path / "w.hdr")
    pkg.hdr_write(path, img)
    mine, ref = pkg.hdr_load(path), R.hdr_load(path)
    assert np.array_equal(mine.view(np.uint32), ref.view(np.uint32))
    # shared-exponent quantisation: at most 2^-8 of the largest channel off
    err = np.abs(mine[..., :3] - img).max(axis=2)
    assert (err <= img.max(axis=2) / 128.0 + 1e-30).all()


@needs_ref
def test_png16_writer_is_read_back_by_the_reference_codec(pkg, tmp_path):
    """A 16-bit RGB PNG this repository writes decodes, in the reference's lodepng, to the samples that went in; and the
    reference's own encoder (state of src/image.cpp:36-43) produces a file with the same header fields."""
    rng = np.random.default_rng(9)
    img = rng.integers(0, 65536, size=(23, 37, 3), dtype=np.uint16)
    img[5:9] = 65535
    mine, theirs = str(tmp_path / "mine.png"), str(tmp_path / "theirs.png")
    pkg.png16_write(mine, img)
    R.png_encode_rgb16(theirs, img)
    assert np.array_equal(R.png_decode_rgb16(mine), img)
    assert np.array_equal(R.png_decode_rgb16(theirs), img)
    # IHDR: width, height, bit depth 16, colour type 2 (RGB), no interlace -- identical in both files
    assert open(mine, "rb").read(33)[:29] == open(theirs, "rb").read(33)[:29]
