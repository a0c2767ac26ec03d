"""ctypes view of the CPU oracle (oracle/liboracle.so).  Test infrastructure only.

The structures here are shared with the product binding: both libraries take the same scene bytes.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

DIFFUSE, MIRROR, GLOSSY, GLASS = 0x4, 0x8, 0x10, 0x20
PRIM_MISS, PRIM_PLANE, PRIM_SPHERE = 0xFFFFFFFF, 0x80000000, 0x40000000


class Plane(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("normal", C.c_float * 3), ("material", C.c_float * 4),
                ("material_type", C.c_uint32)]


class Sphere(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("radius", C.c_float), ("material", C.c_float * 4),
                ("material_type", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("bottom_left", C.c_float * 3), ("right", C.c_float * 3),
                ("up", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3), ("w", C.c_float * 3),
                ("lens_radius", C.c_float)]


class Mesh(C.Structure):
    _fields_ = [("material", C.c_float * 4), ("material_type", C.c_uint32), ("first_face", C.c_uint32)]


class SceneDesc(C.Structure):
    _fields_ = [("bvh_nodes", C.c_void_p), ("num_nodes", C.c_uint64),
                ("bvh_tris", C.c_void_p), ("num_tri_float4", C.c_uint64),
                ("bvh_prim_ids", C.c_void_p), ("num_tris", C.c_uint64),
                ("mesh_material", C.c_float * 4), ("mesh_material_type", C.c_uint32), ("reserved0", C.c_uint32),
                ("planes", C.c_void_p), ("num_planes", C.c_uint64),
                ("spheres", C.c_void_p), ("num_spheres", C.c_uint64),
                ("camera", Camera),
                ("env_width", C.c_uint32), ("env_height", C.c_uint32),
                ("env_texels", C.c_void_p), ("env_marginal", C.c_void_p), ("env_conditional", C.c_void_p),
                ("meshes", C.c_void_p), ("num_meshes", C.c_uint64)]


class Bvh(C.Structure):
    _fields_ = [("nodes", C.POINTER(C.c_float)), ("num_nodes", C.c_uint64),
                ("tris", C.POINTER(C.c_float)), ("num_tri_float4", C.c_uint64),
                ("prim_ids", C.POINTER(C.c_uint32)), ("num_tris", C.c_uint64),
                ("num_leaves", C.c_uint64), ("max_depth", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes", C.c_uint64), ("tris", C.c_uint64), ("leaves", C.c_uint64),
                ("paths", C.c_uint64), ("ms", C.c_double)]


_lib = None


def lib():
    """Loads (building first if the source is newer) oracle/liboracle.so."""
    global _lib
    if _lib is not None:
        return _lib
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    src = os.path.join(ORACLE_DIR, "oracle.cpp")
    if not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    L = C.CDLL(so)
    L.orc_fnv1a64.restype = C.c_uint64
    L.orc_fnv1a64.argtypes = [C.c_void_p, C.c_uint64]
    L.orc_bvh_build.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Bvh)]
    L.orc_bvh_free.argtypes = [C.POINTER(Bvh)]
    L.orc_env_build.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    L.orc_camera_make.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float,
                                  C.POINTER(Camera)]
    L.orc_philox4x32_10.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_intersect.argtypes = [C.POINTER(SceneDesc), C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.POINTER(Stats)]
    L.orc_primary_rays.argtypes = [C.POINTER(SceneDesc), C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64,
                                   C.c_void_p, C.c_void_p]
    L.orc_render.argtypes = [C.POINTER(SceneDesc), C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64,
                             C.c_uint64, C.c_void_p, C.c_int, C.POINTER(Stats)]
    L.orc_tonemap16.argtypes = [C.c_void_p, C.c_uint64, C.c_float, C.c_float, C.c_void_p]
    L.orc_rgbe_to_float.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    L.orc_math.argtypes = [C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_math.restype = None
    _lib = L
    return L


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def fnv1a64(buf: np.ndarray) -> int:
    buf = np.ascontiguousarray(buf)
    return lib().orc_fnv1a64(ptr(buf), buf.nbytes)


def bvh_build(verts: np.ndarray):
    """verts: (ntris, 3, 3) float32 -> dict of numpy copies of the flattened buffers."""
    verts = np.ascontiguousarray(verts, dtype=np.float32)
    out = Bvh()
    rc = lib().orc_bvh_build(ptr(verts), verts.shape[0], C.byref(out))
    if rc != 0:
        raise ValueError("orc_bvh_build failed (%d)" % rc)
    res = {
        "nodes": np.ctypeslib.as_array(out.nodes, shape=(out.num_nodes, 16)).copy(),
        # keep the two float4 of zero padding the leaf loop may touch
        "tris": np.ctypeslib.as_array(out.tris, shape=(out.num_tri_float4 + 2, 4)).copy(),
        "prim_ids": np.ctypeslib.as_array(out.prim_ids, shape=(out.num_tri_float4,)).copy(),
        "num_tri_float4": int(out.num_tri_float4),
        "num_tris": int(out.num_tris),
        "num_leaves": int(out.num_leaves),
        "max_depth": int(out.max_depth),
    }
    lib().orc_bvh_free(C.byref(out))
    return res


def env_build(texels: np.ndarray):
    """texels: (H, W, 4) float32, modified in place (.w <- pdf).  Returns (marginal, conditional) uint64 tables."""
    assert texels.dtype == np.float32 and texels.flags.c_contiguous
    H, W = texels.shape[:2]
    marg = np.zeros(W, dtype=np.uint64)
    cond = np.zeros(W * H, dtype=np.uint64)
    lib().orc_env_build(ptr(texels), W, H, ptr(marg), ptr(cond))
    return marg, cond


def camera_make(pos, look_at, fov, aspect, aperture, focus):
    cam = Camera()
    p = np.asarray(pos, dtype=np.float32)
    l = np.asarray(look_at, dtype=np.float32)
    lib().orc_camera_make(ptr(p), ptr(l), fov, aspect, aperture, focus, C.byref(cam))
    return cam


def philox(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    o = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(ptr(c), ptr(k), ptr(o))
    return o


def intersect(desc, org, dir_, want_normals=False):
    org = np.ascontiguousarray(org, dtype=np.float32)
    dir_ = np.ascontiguousarray(dir_, dtype=np.float32)
    n = org.shape[0]
    t = np.zeros(n, dtype=np.float32)
    prim = np.zeros(n, dtype=np.uint32)
    nrm = np.zeros((n, 3), dtype=np.float32) if want_normals else None
    st = Stats()
    lib().orc_intersect(C.byref(desc), n, ptr(org), ptr(dir_), ptr(t), ptr(prim),
                        ptr(nrm) if want_normals else None, C.byref(st))
    return t, prim, nrm, st


def primary_rays(desc, W, H, sample=0, seed=0):
    org = np.zeros((W * H, 3), dtype=np.float32)
    dir_ = np.zeros((W * H, 3), dtype=np.float32)
    lib().orc_primary_rays(C.byref(desc), W, H, sample, seed, ptr(org), ptr(dir_))
    return org, dir_


def render(desc, W, H, spp_begin, spp_count, spp_total, seed=0, nthreads=None, film=None):
    if film is None:
        film = np.zeros((H, W, 3), dtype=np.float32)
    if nthreads is None:
        nthreads = min(os.cpu_count() or 1, 16)
    st = Stats()
    lib().orc_render(C.byref(desc), W, H, spp_begin, spp_count, spp_total, seed, ptr(film), nthreads, C.byref(st))
    return film, st


def tonemap16(film, exposure=2.0, gamma=2.2):
    film = np.ascontiguousarray(film, dtype=np.float32)
    out = np.zeros(film.shape, dtype=np.uint16)
    lib().orc_tonemap16(ptr(film), film.size // 3, exposure, gamma, ptr(out))
    return out


def math(which, a, b=None):
    """The shading path's fixed transcendental kernels: 0 sincos (two results), 1 atan2(a, b), 2 acos, 3 a^5."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(a if b is None else b, dtype=np.float32)
    out0, out1 = np.zeros_like(a), np.zeros_like(a)
    lib().orc_math(which, a.size, ptr(a), ptr(b), ptr(out0), ptr(out1))
    return (out0, out1) if which == 0 else out0
