"""Freezes what the reference's own .hdr reader (stb_image's stbi_loadf, compiled from /root/reference/include by
`make -C oracle ref`) returns for a fixed set of RGBE pixels: tests/golden/hdr_decode.npz.  Run in the container that has
the reference; the fixture lets the decode be checked where it is absent.

The pixels: every exponent byte with a few mantissa triples, written as a flat (not run-length encoded) Radiance file.
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_io_lib as R  # noqa: E402


def pixels():
    rng = np.random.default_rng(20211)
    px = []
    for e in range(256):
        for m in ((0, 0, 0), (1, 2, 3), (128, 64, 255), (255, 255, 255)):
            px.append(m + (e,))
        px.append(tuple(int(v) for v in rng.integers(0, 256, size=3)) + (e,))
    return np.array(px, dtype=np.uint8)


def write_flat_hdr(path, rgbe, width):
    n = rgbe.shape[0]
    assert n % width == 0
    with open(path, "wb") as fh:
        fh.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (n // width, width))
        fh.write(rgbe.tobytes())


if __name__ == "__main__":
    rgbe = pixels()                      # 1280 pixels
    width = 5                            # below 8: stb_image reads such files flat (stb_image.h:7020-7030)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "flat.hdr")
        write_flat_hdr(path, rgbe, width)
        decoded = R.hdr_load(path)
    out = os.path.join(ROOT, "tests", "golden", "hdr_decode.npz")
    np.savez_compressed(out, rgbe=rgbe, width=np.int32(width), rgba=decoded)
    print("wrote", out, decoded.shape, float(decoded[..., :3].max()))
