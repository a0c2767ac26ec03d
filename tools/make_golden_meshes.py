"""Packs the reference's shipped meshes (data files under /root/reference/obj) into tests/golden/meshes.npz.

Runs only where the reference checkout is mounted.  The .npz holds, per mesh, the float32 vertex table (index 0 is
the dummy vertex the reference's loader inserts, src/scene.cu:20) and the int32 face index triples, i.e. the values
the reference's load_obj (src/scene.cu:15-55) would read; nothing of the reference's source text is stored.
"""
import ctypes
import os
import sys

import numpy as np

REF = "/root/reference/obj"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "meshes.npz")

libc = ctypes.CDLL("libc.so.6")
libc.strtof.restype = ctypes.c_float
libc.strtof.argtypes = [ctypes.c_char_p, ctypes.c_void_p]


def parse(path):
    verts = [(0.0, 0.0, 0.0)]
    faces = []
    for line in open(path):
        s = line.split()
        if not s:
            continue
        if s[0] == "v":
            verts.append(tuple(libc.strtof(x.encode(), None) for x in s[1:4]))
        elif s[0] == "f":
            idx = [int(x) for x in s[1:4]]
            faces.append([i + len(verts) if i < 0 else i for i in idx])
        else:
            raise SystemExit("unexpected record in %s: %r" % (path, line))
    return np.array(verts, dtype=np.float32), np.array(faces, dtype=np.int32)


def main():
    out = {}
    for name in ["bunny_lowpoly", "suzanne", "city", "bunny"]:
        v, f = parse(os.path.join(REF, name + ".obj"))
        out[name + "_v"] = v
        out[name + "_f"] = f
        print(name, v.shape, f.shape)
    np.savez_compressed(OUT, **out)
    print("wrote", os.path.abspath(OUT), os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    sys.exit(main())
