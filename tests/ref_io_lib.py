"""ctypes access to oracle/_ref/libref_io.so: the reference's own vendored .hdr reader (stb_image) and PNG codec (lodepng),
compiled from where they lie under /root/reference (oracle/Makefile, target `ref`).  Test infrastructure only; the
library exists only where the reference does."""
import ctypes as C
import os

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libref_io.so")
_lib = None


def available():
    return os.path.exists(PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(PATH)
        L.ref_hdr_load.restype = C.POINTER(C.c_float)
        L.ref_hdr_load.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ref_free.argtypes = [C.c_void_p]
        L.ref_png_decode_rgb16.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
        L.ref_png_encode_rgb16.argtypes = [C.c_char_p, C.c_void_p, C.c_uint, C.c_uint]
        _lib = L
    return _lib


def hdr_load(path):
    """(H, W, 4) float32 as stbi_loadf(path, .., 4) returns it (reference src/environment_map.cu:24)."""
    w, h = C.c_int(), C.c_int()
    p = lib().ref_hdr_load(os.fsencode(path), C.byref(w), C.byref(h))
    if not p:
        raise RuntimeError("stbi_loadf failed on %s" % path)
    out = np.ctypeslib.as_array(p, shape=(h.value, w.value, 4)).copy()
    lib().ref_free(p)
    return out


def png_decode_rgb16(path):
    """(H, W, 3) uint16 samples of a PNG, decoded by the reference's lodepng."""
    out = C.POINTER(C.c_ubyte)()
    w, h = C.c_uint(), C.c_uint()
    err = lib().ref_png_decode_rgb16(os.fsencode(path), C.byref(out), C.byref(w), C.byref(h))
    if err:
        raise RuntimeError("lodepng error %d on %s" % (err, path))
    raw = np.ctypeslib.as_array(out, shape=(h.value, w.value, 3, 2)).copy()
    lib().ref_free(out)
    return (raw[..., 0].astype(np.uint16) << 8) | raw[..., 1]


def png_encode_rgb16(path, rgb16):
    """Writes (H, W, 3) uint16 samples with the encoder state of reference src/image.cpp:36-43."""
    be = np.ascontiguousarray(rgb16.astype(">u2"))
    err = lib().ref_png_encode_rgb16(os.fsencode(path), be.ctypes.data_as(C.c_void_p), rgb16.shape[1], rgb16.shape[0])
    if err:
        raise RuntimeError("lodepng error %d writing %s" % (err, path))
