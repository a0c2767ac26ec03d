"""Python host-side mirror of the path tracer's C ABI (include/pt_api.h, include/pt_host.h).

Thin ctypes layer over the two in-tree shared libraries:
  libpt_host.so  scene assembly on the CPU (.scene/.obj/.hdr readers, BVH build, environment tables, PNG output)
  libpt_hip.so   the gfx950 hot path (closest-hit traversal, wavefront render loop)
There is no CPU fallback for the device path: creating a Context without the HIP library or without a GPU raises.
"""
import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))

DIFFUSE, MIRROR, GLOSSY, GLASS = 0x4, 0x8, 0x10, 0x20
PRIM_MISS, PRIM_PLANE, PRIM_SPHERE = 0xFFFFFFFF, 0x80000000, 0x40000000
OPT_POOL_SIZE, OPT_COUNT_TRAVERSAL, OPT_TIME_EXTEND, OPT_SLAB_BYTES, OPT_EXTEND_MODE = 1, 2, 3, 4, 5
OPT_TUNE_REFILL_IDLE, OPT_TUNE_LDS_SLOTS = 6, 7
OPT_ENV_PACK = 8
OPT_STREAMS = 9
OPT_TUNE_EXTEND_BLOCKS = 10
OPT_NODE_LAYOUT = 11
OPT_TUNE_XCD_GROUP = 12
OPT_RENDER_MODE, OPT_TUNE_PATH_WAVES, OPT_TUNE_PATH_REFILL, OPT_TUNE_PATH_LDS_SLOTS = 13, 14, 15, 16


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


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("paths", C.c_uint64), ("iterations", C.c_uint64),
                ("extend_launches", C.c_uint64), ("ms_total", C.c_double), ("ms_extend", C.c_double),
                ("nodes", C.c_uint64), ("tris", C.c_uint64), ("leaves", C.c_uint64),
                ("ms_extend_busy", C.c_double), ("rays_escaped", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Bvh(C.Structure):
    _fields_ = [("nodes", C.POINTER(C.c_float)), ("num_nodes", C.c_uint64),
                ("tris", C.POINTER(C.c_float)), ("num_tri_float4", C.c_uint64),
                ("prim_ids", C.POINTER(C.c_uint32)), ("num_tris", C.c_uint64),
                ("num_leaves", C.c_uint64), ("max_depth", C.c_uint32)]


class PtError(RuntimeError):
    pass


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


_host = None
_hip = None


def host_lib():
    global _host
    if _host is None:
        path = os.path.join(PKG_DIR, "libpt_host.so")
        if not os.path.exists(path):
            raise PtError("libpt_host.so is not built: run `make -C %s`" % PKG_DIR)
        L = C.CDLL(path)
        L.pth_last_error.restype = C.c_char_p
        L.pth_scene_load.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        L.pth_scene_free.argtypes = [C.c_void_p]
        L.pth_scene_desc.restype = C.POINTER(SceneDesc)
        L.pth_scene_desc.argtypes = [C.c_void_p]
        L.pth_obj_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint64)]
        L.pth_free.argtypes = [C.c_void_p]
        L.pth_bvh_build.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(Bvh)]
        L.pth_bvh_free.argtypes = [C.POINTER(Bvh)]
        L.pth_hdr_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint32),
                                   C.POINTER(C.c_uint32)]
        L.pth_hdr_write.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
        L.pth_env_build.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.pth_camera_make.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float,
                                      C.POINTER(Camera)]
        L.pth_tonemap16.argtypes = [C.c_void_p, C.c_uint64, C.c_float, C.c_float, C.c_void_p]
        L.pth_png16_write.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.pth_pfm_write.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.pth_asset_generate.argtypes = [C.c_char_p, C.c_char_p]
        _host = L
    return _host


def hip_lib():
    """The device library.  Raises if it is missing: there is no fallback path."""
    global _hip
    if _hip is None:
        # PT_HIP_LIB: another build of the library (A/B runs of compile-time variants)
        path = os.environ.get("PT_HIP_LIB") or os.path.join(PKG_DIR, "libpt_hip.so")
        if not os.path.exists(path):
            raise PtError("libpt_hip.so is not built: run `make -C %s`" % PKG_DIR)
        try:
            # PyTorch wheels bundle their own HIP/HSA runtime under the same soname.  Two runtimes in one process do
            # not share the GPU (the second reports "no HIP GPUs"), so when torch is installed it is loaded first and
            # this library binds to the runtime torch brought.
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(path)
        L.pt_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.pt_ctx_destroy.argtypes = [C.c_void_p]
        L.pt_last_error.restype = C.c_char_p
        L.pt_last_error.argtypes = [C.c_void_p]
        L.pt_set_option.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
        L.pt_scene_upload.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
        L.pt_intersect.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pt_intersect_device.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.POINTER(Stats)]
        L.pt_intersect_device_noalloc.argtypes = L.pt_intersect_device.argtypes
        L.pt_primary_rays.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p,
                                      C.c_void_p]
        L.pt_math_eval.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pt_film_bind.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.pt_film_clear.argtypes = [C.c_void_p]
        L.pt_film_download.argtypes = [C.c_void_p, C.c_void_p]
        L.pt_film_tonemap16.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.pt_film_reduce.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int]
        L.pt_render.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(Stats)]
        L.pt_env_is_packed.argtypes = [C.c_void_p]
        L.pt_render_mode_used.argtypes = [C.c_void_p]
        L.pt_stream.restype = C.c_void_p
        L.pt_stream.argtypes = [C.c_void_p]
        _hip = L
    return _hip


def _hcheck(rc):
    if rc != 0:
        raise PtError(host_lib().pth_last_error().decode())


# ---- host side ---------------------------------------------------------------------------------------------------
class HostScene:
    """A scene assembled on the host: the counterpart of the reference's Scene constructor minus the uploads."""

    def __init__(self, scene_file, width, height):
        self._h = None
        h = C.c_void_p()
        _hcheck(host_lib().pth_scene_load(os.fsencode(scene_file), width, height, C.byref(h)))
        self._h = h
        self.width, self.height = width, height
        self.desc = host_lib().pth_scene_desc(h).contents

    def close(self):
        if getattr(self, "_h", None):
            host_lib().pth_scene_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


class ManualScene:
    """A scene built from numpy buffers (planes, spheres, BVH buffers, environment) rather than from a file."""

    def __init__(self, camera, env_texels, env_marginal, env_conditional, planes=(), spheres=(), bvh=None,
                 mesh_material=(0.7, 0.7, 0.7, 0.0), mesh_material_type=DIFFUSE, meshes=None):
        """meshes: optional list of (material, material_type, first_face): several meshes in the one BVH, each with its
        own material (a triangle belongs to the last mesh whose first_face <= its prim id)."""
        d = SceneDesc()
        self._keep = []
        if bvh is not None:
            nodes = np.ascontiguousarray(bvh["nodes"], dtype=np.float32)
            tris = np.ascontiguousarray(bvh["tris"], dtype=np.float32)
            ids = np.ascontiguousarray(bvh["prim_ids"], dtype=np.uint32)
            self._keep += [nodes, tris, ids]
            d.bvh_nodes, d.num_nodes = nodes.ctypes.data, nodes.shape[0]
            d.bvh_tris, d.num_tri_float4 = tris.ctypes.data, int(bvh["num_tri_float4"])
            d.bvh_prim_ids, d.num_tris = ids.ctypes.data, int(bvh["num_tris"])
        d.mesh_material = (C.c_float * 4)(*mesh_material)
        d.mesh_material_type = mesh_material_type
        if meshes:
            arr = (Mesh * len(meshes))(*[Mesh((C.c_float * 4)(*m), t, f) for m, t, f in meshes])
            self._keep.append(arr)
            d.meshes, d.num_meshes = C.addressof(arr), len(meshes)
        if planes:
            arr = (Plane * len(planes))(*planes)
            self._keep.append(arr)
            d.planes, d.num_planes = C.addressof(arr), len(planes)
        if spheres:
            arr = (Sphere * len(spheres))(*spheres)
            self._keep.append(arr)
            d.spheres, d.num_spheres = C.addressof(arr), len(spheres)
        d.camera = camera
        tex = np.ascontiguousarray(env_texels, dtype=np.float32)
        marg = np.ascontiguousarray(env_marginal, dtype=np.uint64)
        cond = np.ascontiguousarray(env_conditional, dtype=np.uint64)
        self._keep += [tex, marg, cond]
        d.env_height, d.env_width = tex.shape[0], tex.shape[1]
        d.env_texels, d.env_marginal, d.env_conditional = tex.ctypes.data, marg.ctypes.data, cond.ctypes.data
        self.desc = d


def make_plane(position, normal, material, material_type):
    n = np.asarray(normal, dtype=np.float32)
    n = n * (np.float32(1.0) / np.sqrt(np.float32(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), dtype=np.float32))
    return Plane((C.c_float * 3)(*position), (C.c_float * 3)(*n), (C.c_float * 4)(*material), material_type)


def make_sphere(position, radius, material, material_type):
    return Sphere((C.c_float * 3)(*position), radius, (C.c_float * 4)(*material), material_type)


def obj_load(path):
    v = C.POINTER(C.c_float)()
    n = C.c_uint64()
    _hcheck(host_lib().pth_obj_load(os.fsencode(path), C.byref(v), C.byref(n)))
    out = np.ctypeslib.as_array(v, shape=(n.value, 3, 3)).copy() if n.value else np.zeros((0, 3, 3), np.float32)
    host_lib().pth_free(v)
    return out


def bvh_build(verts, threads=1):
    verts = np.ascontiguousarray(verts, dtype=np.float32)
    out = Bvh()
    _hcheck(host_lib().pth_bvh_build(_ptr(verts), verts.shape[0], threads, C.byref(out)))
    res = {
        "nodes": np.ctypeslib.as_array(out.nodes, shape=(out.num_nodes, 16)).copy(),
        "tris": np.ctypeslib.as_array(out.tris, shape=(out.num_tri_float4 + 2, 4)).copy(),
        "prim_ids": np.ctypeslib.as_array(out.prim_ids, shape=(out.num_tri_float4,)).copy(),
        "num_tri_float4": int(out.num_tri_float4),
        "num_tris": int(out.num_tris),
        "num_leaves": int(out.num_leaves),
        "max_depth": int(out.max_depth),
    }
    host_lib().pth_bvh_free(C.byref(out))
    return res


def hdr_load(path):
    p = C.POINTER(C.c_float)()
    w, h = C.c_uint32(), C.c_uint32()
    _hcheck(host_lib().pth_hdr_load(os.fsencode(path), C.byref(p), C.byref(w), C.byref(h)))
    out = np.ctypeslib.as_array(p, shape=(h.value, w.value, 4)).copy()
    host_lib().pth_free(p)
    return out


def hdr_write(path, rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    _hcheck(host_lib().pth_hdr_write(os.fsencode(path), _ptr(rgb), rgb.shape[1], rgb.shape[0], rgb.shape[2]))


def env_build(texels):
    assert texels.dtype == np.float32 and texels.flags.c_contiguous and texels.shape[2] == 4
    H, W = texels.shape[:2]
    marg = np.zeros(W, dtype=np.uint64)
    cond = np.zeros(W * H, dtype=np.uint64)
    _hcheck(host_lib().pth_env_build(_ptr(texels), W, H, _ptr(marg), _ptr(cond)))
    return marg, cond


def camera_make(pos, look_at, fov, aspect, aperture, focus):
    cam = Camera()
    p = np.asarray(pos, dtype=np.float32)
    l = np.asarray(look_at, dtype=np.float32)
    host_lib().pth_camera_make(_ptr(p), _ptr(l), fov, aspect, aperture, focus, C.byref(cam))
    return cam


def tonemap16(film, exposure=2.0, gamma=2.2):
    film = np.ascontiguousarray(film, dtype=np.float32)
    out = np.zeros(film.shape, dtype=np.uint16)
    host_lib().pth_tonemap16(_ptr(film), film.size // 3, exposure, gamma, _ptr(out))
    return out


def png16_write(path, rgb16):
    rgb16 = np.ascontiguousarray(rgb16, dtype=np.uint16)
    _hcheck(host_lib().pth_png16_write(os.fsencode(path), _ptr(rgb16), rgb16.shape[1], rgb16.shape[0]))


def pfm_write(path, film):
    film = np.ascontiguousarray(film, dtype=np.float32)
    _hcheck(host_lib().pth_pfm_write(os.fsencode(path), _ptr(film), film.shape[1], film.shape[0]))


def asset_generate(kind, path):
    _hcheck(host_lib().pth_asset_generate(kind.encode(), os.fsencode(path)))


# ---- device side -------------------------------------------------------------------------------------------------
class Context:
    """One device context: the counterpart of the reference's Scene device state plus render_scene."""

    def __init__(self, device=0):
        self._L = hip_lib()
        h = C.c_void_p()
        rc = self._L.pt_ctx_create(device, C.byref(h))
        if rc != 0:
            raise PtError("pt_ctx_create: %s" % self._L.pt_last_error(None).decode())
        self._h = h
        self.film_shape = None

    def _check(self, rc):
        if rc != 0:
            raise PtError(self._L.pt_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.pt_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def set_option(self, option, value):
        self._check(self._L.pt_set_option(self._h, option, int(value)))

    def upload(self, scene):
        desc = scene.desc if hasattr(scene, "desc") else scene
        self._check(self._L.pt_scene_upload(self._h, C.byref(desc)))

    def intersect(self, org, dir_, want_normals=False):
        org = np.ascontiguousarray(org, dtype=np.float32)
        dir_ = np.ascontiguousarray(dir_, dtype=np.float32)
        n = org.shape[0]
        t = np.zeros(n, dtype=np.float32)
        prim = np.zeros(n, dtype=np.uint32)
        nrm = np.zeros((n, 3), dtype=np.float32) if want_normals else None
        self._check(self._L.pt_intersect(self._h, n, _ptr(org), _ptr(dir_), _ptr(t), _ptr(prim),
                                         _ptr(nrm) if want_normals else None))
        return t, prim, nrm

    def intersect_device(self, n, d_org4, d_dir4, d_hit4, d_t):
        """Device pointers (ints).  Returns Stats."""
        st = Stats()
        self._check(self._L.pt_intersect_device(self._h, n, d_org4, d_dir4, d_hit4, d_t, C.byref(st)))
        return st

    def intersect_device_noalloc(self, n, d_org4, d_dir4, d_hit4, d_t, want_stats=True):
        """intersect_device without any device allocation (static scratch, at most 4,194,304 rays).  Returns Stats; with
        want_stats=False the kernels are only queued on the context's stream (no synchronisation) and None is returned."""
        st = Stats() if want_stats else None
        self._check(self._L.pt_intersect_device_noalloc(self._h, n, d_org4, d_dir4, d_hit4, d_t,
                                                        C.byref(st) if want_stats else None))
        return st

    def primary_rays(self, width, height, sample=0, seed=0):
        org = np.zeros((width * height, 3), dtype=np.float32)
        dir_ = np.zeros((width * height, 3), dtype=np.float32)
        self._check(self._L.pt_primary_rays(self._h, width, height, sample, seed, _ptr(org), _ptr(dir_)))
        return org, dir_

    def math_eval(self, which, a, b=None):
        """The device's evaluation of the shading path's transcendental kernels: 0 sincos (two results), 1 atan2(a, b),
        2 acos, 3 a^5."""
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(a if b is None else b, dtype=np.float32)
        out0, out1 = np.zeros_like(a), np.zeros_like(a)
        self._check(self._L.pt_math_eval(self._h, which, a.size, _ptr(a), _ptr(b), _ptr(out0), _ptr(out1)))
        return (out0, out1) if which == 0 else out0

    def film_bind(self, width, height, device_ptr=None):
        self._check(self._L.pt_film_bind(self._h, width, height, device_ptr))
        self.film_shape = (height, width, 3)

    def film_clear(self):
        self._check(self._L.pt_film_clear(self._h))

    def film_download(self):
        out = np.zeros(self.film_shape, dtype=np.float32)
        self._check(self._L.pt_film_download(self._h, _ptr(out)))
        return out

    def film_tonemap16(self, exposure=2.0, gamma=2.2):
        out = np.zeros(self.film_shape, dtype=np.uint16)
        self._check(self._L.pt_film_tonemap16(self._h, exposure, gamma, _ptr(out)))
        return out

    def render(self, spp_begin, spp_count, spp_total, seed=0):
        st = Stats()
        self._check(self._L.pt_render(self._h, spp_begin, spp_count, spp_total, seed, C.byref(st)))
        return st

    @property
    def render_mode_used(self):
        """1 wavefront loop, 2 whole-path kernel: what the last render ran in (0 before the first)."""
        return self._L.pt_render_mode_used(self._h)

    @property
    def stream(self):
        return self._L.pt_stream(self._h)

    @property
    def env_is_packed(self):
        """True when the uploaded scene keeps its environment texels in the packed 8-byte form."""
        rc = self._L.pt_env_is_packed(self._h)
        if rc < 0:
            raise PtError("no scene uploaded")
        return rc == 1


def film_reduce(contexts, root=0):
    """Adds the films of the other contexts to contexts[root]'s (one process driving several GPUs)."""
    arr = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    rc = hip_lib().pt_film_reduce(arr, len(contexts), root)
    if rc != 0:
        raise PtError("pt_film_reduce: %s" % hip_lib().pt_last_error(contexts[root]._h).decode())


def spp_partition(spp_total, world_size, rank):
    """Samples-per-pixel range [begin, begin+count) of `rank` when spp_total is split over world_size ranks."""
    begin = spp_total * rank // world_size
    end = spp_total * (rank + 1) // world_size
    return begin, end - begin
