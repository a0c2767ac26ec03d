"""Scenes shared by the tests, the smoke check and the benchmark.

Large generated assets (environment maps, the dragon stand-in) are written once to a cache directory outside the
repository ($PT_ASSET_DIR, default /tmp/pt_assets).  The .scene texts below carry the reference's own camera,
material and primitive values (reference scene/*.scene); only the asset paths differ.
"""
import ctypes as C
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def asset_dir():
    d = os.environ.get("PT_ASSET_DIR", "/tmp/pt_assets")
    os.makedirs(d, exist_ok=True)
    return d


def ensure_asset(pkg, kind):
    path = os.path.join(asset_dir(), kind)
    if not os.path.exists(path):
        tmp = path + ".tmp%d" % os.getpid()
        pkg.asset_generate(kind, tmp)
        os.replace(tmp, path)
    return path


_MESHES = None


def golden_mesh(name):
    """(ntris, 3, 3) float32 triangles of one of the reference's shipped meshes, from tests/golden/meshes.npz."""
    global _MESHES
    if _MESHES is None:
        _MESHES = np.load(os.path.join(GOLDEN, "meshes.npz"))
    v = _MESHES[name + "_v"]
    f = _MESHES[name + "_f"]
    return np.ascontiguousarray(v[f], dtype=np.float32)


def write_obj(path, tris):
    """Writes triangles as an OBJ with the exact float values (%.9g round-trips float32)."""
    with open(path, "w") as fh:
        flat = tris.reshape(-1, 3)
        for p in flat:
            fh.write("v %.9g %.9g %.9g\n" % (p[0], p[1], p[2]))
        for i in range(tris.shape[0]):
            fh.write("f %d %d %d\n" % (3 * i + 1, 3 * i + 2, 3 * i + 3))


def ensure_mesh_obj(name):
    path = os.path.join(asset_dir(), name + ".obj")
    if not os.path.exists(path):
        tmp = path + ".tmp%d" % os.getpid()
        write_obj(tmp, golden_mesh(name))
        os.replace(tmp, path)
    return path


SCENE_TEXT = {
    # reference scene/balls.scene
    "balls": """CAMERA
position 0 1 5
look_at 0 0.5 0
fov 35
aperture 0.1
focus_dist 4.75

ENVIRONMENT
file_name {studio}

PLANE
material_type LAMBERTIAN
albedo 0.325 0.3 0.35
position 0 0 0
normal 0 1 0

SPHERE
material_type GLASS
albedo 1 1 1
refraction_index 1.5
position -2.025 0.5 0
radius 0.5

SPHERE
material_type LAMBERTIAN
albedo 0.7 0.7 0.7
position -0.675 0.5 0
radius 0.5

SPHERE
material_type MIRROR
albedo 0.5 0.5 0.5
position 0.675 0.5 0
radius 0.5

SPHERE
material_type GLOSSY
albedo 0.15 0.25 0.4
specular 1 1 1
glossyness 0.05
position 2.025 0.5 0
radius 0.5
""",
    # reference scene/bunny.scene
    "bunny": """CAMERA
position 1 1 4
look_at -1 0.5 0
fov 35
aperture 0.1
focus_dist 4

ENVIRONMENT
file_name {sunset}

PLANE
material_type LAMBERTIAN
albedo 0.08 0.08 0.12
position 0 0 0
normal 0 1 0

SPHERE
material_type MIRROR
albedo 0.5 0.5 0.5
position -3 1.5 -0.5
radius 1.5

OBJ
material_type LAMBERTIAN
albedo 0.6 0.3 0.1
file_name {bunny}
""",
    # reference scene/dragon.scene (the mesh is the procedural stand-in)
    "dragon": """CAMERA
position -1.1 0.2 -0.8
look_at 0 0.0 0
fov 38
aperture 0.01
focus_dist 1.15

ENVIRONMENT
file_name {forest}

PLANE
material_type LAMBERTIAN
albedo 0.65 0.1 0.1
position 0 -0.283 0
normal 0 1 0

OBJ
material_type GLOSSY
albedo 0.65 0.65 0.65
specular 1 1 1
glossyness 0.05
file_name {dragon}
""",
    # reference scene/city.scene
    "city": """CAMERA
position -10 12 10
look_at 0 2 0
fov 38
aperture 0.05
focus_dist 15

ENVIRONMENT
file_name {studio}

PLANE
material_type LAMBERTIAN
albedo 0.1 0.08 0.15
position 0 0 0
normal 0 1 0

OBJ
material_type LAMBERTIAN
albedo 0.7 0.7 0.7
file_name {city}
""",
}

# the dragon scene with the frame-filling variant of the stand-in mesh (91 % of the camera rays hit it, 44 inner nodes per
# ray): the traversal-bound workload
SCENE_TEXT["dragon_dense"] = SCENE_TEXT["dragon"].replace("{dragon}", "{dragon_dense}")

# reference scene/suzanne.scene: a pinhole camera (aperture 0), a plane facing +z behind the mesh
SCENE_TEXT["suzanne"] = """CAMERA
position 0 -0.05 4
look_at 0 -0.05 0
fov 35
aperture 0
focus_dist 3.5

ENVIRONMENT
file_name {studio}

PLANE
material_type LAMBERTIAN
albedo 0.06 0.12 0.25
position 0 0 -0.86
normal 0 0 1

OBJ
material_type LAMBERTIAN
albedo 0.2 0.89 0.915
file_name {suzanne}
"""

# two OBJ blocks with different materials (a superset of the reference's one-OBJ scenes): the shipped low-poly bunny,
# Lambertian, and the shipped suzanne moved to its side, glass
SCENE_TEXT["two_meshes"] = """CAMERA
position 0.5 1.6 5.5
look_at 0.8 0.9 0
fov 35
aperture 0.02
focus_dist 5.5

ENVIRONMENT
file_name {studio}

PLANE
material_type LAMBERTIAN
albedo 0.3 0.3 0.35
position 0 0 0
normal 0 1 0

OBJ
material_type LAMBERTIAN
albedo 0.6 0.3 0.1
file_name {bunny_lowpoly}

OBJ
material_type GLASS
albedo 0.9 1 0.9
refraction_index 1.45
file_name {suzanne_moved}

SPHERE
material_type MIRROR
albedo 0.6 0.6 0.6
position -1.6 0.5 0.3
radius 0.5
"""

_SCENE_ASSETS = {"balls": ["studio.hdr"], "bunny": ["sunset.hdr"], "dragon": ["forest.hdr", "dragon.obj"],
                 "dragon_dense": ["forest.hdr", "dragon_dense.obj"], "city": ["studio.hdr"], "two_meshes": ["studio.hdr"], "suzanne": ["studio.hdr"]}


def scene_file(pkg, name):
    """Writes <asset_dir>/<name>.scene (generating what it references) and returns its path."""
    paths = {}
    for kind in _SCENE_ASSETS[name]:
        paths[kind.split(".")[0]] = ensure_asset(pkg, kind)
    if name in ("bunny", "city", "suzanne"):
        paths[name] = ensure_mesh_obj(name)
    if name == "two_meshes":
        paths["bunny_lowpoly"] = ensure_mesh_obj("bunny_lowpoly")
        paths["suzanne_moved"] = os.path.join(asset_dir(), "suzanne_moved.obj")
        if not os.path.exists(paths["suzanne_moved"]):
            tmp = paths["suzanne_moved"] + ".tmp%d" % os.getpid()
            write_obj(tmp, two_meshes_triangles()[1])
            os.replace(tmp, tmp[:tmp.index(".tmp")])
    path = os.path.join(asset_dir(), name + ".scene")
    text = SCENE_TEXT[name].format(**paths)
    if not os.path.exists(path) or open(path).read() != text:
        tmp = path + ".tmp%d" % os.getpid()
        with open(tmp, "w") as fh:
            fh.write(text)
        os.replace(tmp, path)
    return path


def two_meshes_triangles():
    """The triangles of the two_meshes scene's OBJ blocks, in block order."""
    a = golden_mesh("bunny_lowpoly")
    b = (golden_mesh("suzanne") * np.float32(0.8) + np.array([1.9, 0.9, 0.2], dtype=np.float32)).astype(np.float32)
    return a, b


def small_env(pkg, width=64, height=32):
    """A small closed-form lat-long map: dim gradient, one bright patch, one dark band (zero-luminance texels)."""
    j, i = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
    u = (i + 0.5) / width
    v = (j + 0.5) / height
    tex = np.zeros((height, width, 4), dtype=np.float32)
    tex[..., 0] = 0.3 + 0.5 * v
    tex[..., 1] = 0.4 + 0.2 * u
    tex[..., 2] = 0.8 - 0.5 * v
    patch = (np.abs(u - 0.3) < 0.06) & (np.abs(v - 0.3) < 0.08)
    tex[patch, :3] = np.array([40.0, 35.0, 25.0], dtype=np.float32)
    tex[(v > 0.9), :3] = 0.0
    tex[..., 3] = 1.0
    marg, cond = pkg.env_build(tex)
    return tex, marg, cond


def knot_mesh(pkg):
    return pkg.obj_load(ensure_asset(pkg, "knot_small.obj"))


def knot_scene(pkg, width, height, mesh_type=None, with_spheres=True):
    """9,216-triangle knot (glossy) on a plane with glass, mirror and diffuse spheres around it; small environment."""
    tris = knot_mesh(pkg)
    bvh = pkg.bvh_build(tris)
    cam = pkg.camera_make((-1.1, 0.35, -0.9), (0, -0.05, 0), 40.0, width / height, 0.02, 1.3)
    tex, marg, cond = small_env(pkg)
    planes = [pkg.make_plane((0, -0.283, 0), (0, 1, 0), (0.65, 0.1, 0.1, 0.0), pkg.DIFFUSE)]
    spheres = []
    if with_spheres:
        spheres = [
            pkg.make_sphere((0.1, -0.133, -0.62), 0.15, (1.0, 1.0, 1.0, 1.5), pkg.GLASS),
            pkg.make_sphere((-0.55, -0.163, 0.1), 0.12, (0.5, 0.5, 0.5, 0.0), pkg.MIRROR),
            pkg.make_sphere((0.55, -0.183, -0.25), 0.10, (0.7, 0.7, 0.2, 0.0), pkg.DIFFUSE),
        ]
    return pkg.ManualScene(cam, tex, marg, cond, planes=planes, spheres=spheres, bvh=bvh,
                           mesh_material=(0.65, 0.65, 0.65, 0.05),
                           mesh_material_type=pkg.GLOSSY if mesh_type is None else mesh_type)


_INSTANCED = {}


def instanced_scene(pkg, width, height, instances, base="dragon.obj"):
    """The stress configuration: `instances` copies of the dragon stand-in (868,352 triangles each; 116 copies are
    100.7 M triangles) on a square grid in ONE BVH, the copies alternately glossy and glass (a mesh each, see
    pt_scene_desc.meshes), glass spheres between them, the dragon scene's plane and the forest map.
    Returns (scene, triangle count)."""
    import math
    side = int(math.ceil(math.sqrt(instances)))
    key = (instances, base)
    if key not in _INSTANCED:
        base_tris = pkg.obj_load(ensure_asset(pkg, base))          # (n, 3, 3)
        lo = base_tris.reshape(-1, 3).min(axis=0)
        hi = base_tris.reshape(-1, 3).max(axis=0)
        pitch = 1.15 * float(max(hi[0] - lo[0], hi[2] - lo[2]))
        n = base_tris.shape[0]
        tris = np.empty((instances * n, 3, 3), dtype=np.float32)
        centres = []
        for k in range(instances):
            gx, gz = k % side, k // side
            off = np.array([(gx - 0.5 * (side - 1)) * pitch, 0.0, (gz - 0.5 * (side - 1)) * pitch], dtype=np.float32)
            np.add(base_tris, off, out=tris[k * n:(k + 1) * n])
            if gx + 1 < side and gz + 1 < side and len(centres) < 64:
                centres.append((float(off[0] + 0.5 * pitch), float(lo[1]) + 0.12, float(off[2] + 0.5 * pitch)))
        bvh = pkg.bvh_build(tris, threads=min(16, os.cpu_count() or 1))
        ntris = tris.shape[0]
        del tris
        _INSTANCED.clear()                     # one mesh of this size at a time
        _INSTANCED[key] = (bvh, ntris, centres, float(lo[1]), pitch)
    bvh, ntris, centres, floor_y, pitch = _INSTANCED[key]
    spheres = [pkg.make_sphere(c, 0.12, (1.0, 1.0, 1.0, 1.5), pkg.GLASS) for c in centres]
    extent = 0.5 * side * pitch
    cam = pkg.camera_make((-1.6 * extent - 0.8, 0.9 * extent + 0.3, -1.3 * extent - 0.6), (0.0, 0.0, 0.0), 38.0,
                          width / height, 0.01, 2.2 * extent + 1.0)
    tex, marg, cond = hdr_env(pkg, "forest.hdr")
    planes = [pkg.make_plane((0, floor_y, 0), (0, 1, 0), (0.65, 0.1, 0.1, 0.0), pkg.DIFFUSE)]
    # every copy is a mesh of its own in the one BVH: glossy and glass alternate
    n = ntris // instances
    meshes = [((0.65, 0.65, 0.65, 0.05), pkg.GLOSSY, k * n) if k % 2 == 0 else ((0.9, 1.0, 0.9, 1.5), pkg.GLASS, k * n)
              for k in range(instances)]
    scene = pkg.ManualScene(cam, tex, marg, cond, planes=planes, spheres=spheres, bvh=bvh,
                            mesh_material=(0.65, 0.65, 0.65, 0.05), mesh_material_type=pkg.GLOSSY, meshes=meshes)
    return scene, ntris


def hdr_env(pkg, kind):
    """(texels with pdf, marginal table, conditional table) of one of the generated .hdr maps."""
    rgb = pkg.hdr_load(ensure_asset(pkg, kind))
    tex = np.zeros(rgb.shape[:2] + (4,), dtype=np.float32)
    tex[..., :3] = rgb[..., :3]
    marg, cond = pkg.env_build(tex)
    return tex, marg, cond


def mesh_scene(pkg, tris, width, height, cam_pos, look_at, fov=35.0, aperture=0.0, focus=4.0,
               mesh_type=None, planes=None, spheres=None):
    bvh = pkg.bvh_build(tris)
    cam = pkg.camera_make(cam_pos, look_at, fov, width / height, aperture, focus)
    tex, marg, cond = small_env(pkg)
    return pkg.ManualScene(cam, tex, marg, cond, planes=planes or [], spheres=spheres or [], bvh=bvh,
                           mesh_material=(0.6, 0.3, 0.1, 0.0),
                           mesh_material_type=pkg.DIFFUSE if mesh_type is None else mesh_type)


def spheres_scene(pkg, width, height):
    """The reference's balls scene (one plane, four spheres, all four materials) over the small environment."""
    cam = pkg.camera_make((0, 1, 5), (0, 0.5, 0), 35.0, width / height, 0.1, 4.75)
    tex, marg, cond = small_env(pkg)
    planes = [pkg.make_plane((0, 0, 0), (0, 1, 0), (0.325, 0.3, 0.35, 0.0), pkg.DIFFUSE)]
    spheres = [
        pkg.make_sphere((-2.025, 0.5, 0), 0.5, (1, 1, 1, 1.5), pkg.GLASS),
        pkg.make_sphere((-0.675, 0.5, 0), 0.5, (0.7, 0.7, 0.7, 0.0), pkg.DIFFUSE),
        pkg.make_sphere((0.675, 0.5, 0), 0.5, (0.5, 0.5, 0.5, 0.0), pkg.MIRROR),
        pkg.make_sphere((2.025, 0.5, 0), 0.5, (0.15, 0.25, 0.4, 0.05), pkg.GLOSSY),
    ]
    return pkg.ManualScene(cam, tex, marg, cond, planes=planes, spheres=spheres)


def chain_scene(pkg, levels, width=48, height=32):
    """A degenerate BVH of `levels` inner nodes in a chain, written by hand in the reference's buffer layout
    (src/bvh.cu:36-41, 224-226, 275-277): node i has the one-triangle leaf of triangle i on the left and the rest of the
    chain on the right (the last node has two leaves).  Triangle i is a slab at x = i, so a ray travelling towards -x from
    beyond the last triangle finds the rest-of-chain box nearer at every node, pushes the leaf and descends: its traversal
    stack grows to the full depth of the tree.  Woop rows come from the product's builder (a throw-away tree over the
    same triangles)."""
    K = levels + 1
    tris = np.zeros((K, 3, 3), dtype=np.float32)
    for k in range(K):
        tris[k] = [[k, -1.0 - 0.01 * k, -1.0], [k, 1.0, -1.0 - 0.01 * k], [k, 0.0, 1.0 + 0.01 * k]]
    ref = pkg.bvh_build(tris)
    # prim_ids names a triangle at the float4 index of its first row only: keep the smallest index per triangle
    first_row = {}
    a = 1
    while a < ref["num_tri_float4"]:
        if ref["tris"][a].view(np.uint32)[0] == 0x80000000:
            a += 1
            continue
        first_row[int(ref["prim_ids"][a])] = a
        a += 3
    assert sorted(first_row) == list(range(K))
    buf = np.zeros((1 + 4 * K + 2, 4), dtype=np.float32)
    ids = np.zeros(1 + 4 * K, dtype=np.uint32)
    for k in range(K):
        at = 1 + 4 * k
        buf[at:at + 3] = ref["tris"][first_row[k]:first_row[k] + 3]
        buf[at + 3].view(np.uint32)[:] = 0x80000000
        ids[at] = k
    lo = tris.min(axis=1) - np.float32(0.001)
    hi = tris.max(axis=1) + np.float32(0.001)
    nodes = np.zeros((levels, 16), dtype=np.float32)
    for i in range(levels):
        rlo, rhi = lo[i + 1:].min(axis=0), hi[i + 1:].max(axis=0)
        nodes[i, 0:4] = [lo[i, 0], hi[i, 0], lo[i, 1], hi[i, 1]]
        nodes[i, 4:8] = [rlo[0], rhi[0], rlo[1], rhi[1]]
        nodes[i, 8:12] = [lo[i, 2], hi[i, 2], rlo[2], rhi[2]]
        ch = nodes[i].view(np.int32)
        ch[12] = ~np.int32(1 + 4 * i)
        ch[13] = i + 1 if i + 1 < levels else ~np.int32(1 + 4 * (i + 1))
    bvh = {"nodes": nodes, "tris": buf, "prim_ids": ids, "num_tri_float4": 1 + 4 * K, "num_tris": K}
    cam = pkg.camera_make((K + 3.0, 0.2, 0.3), (0, 0, 0), 30.0, width / height, 0.0, 4.0)
    tex, marg, cond = small_env(pkg)
    return pkg.ManualScene(cam, tex, marg, cond, bvh=bvh, mesh_material=(0.6, 0.6, 0.6, 0.0),
                           mesh_material_type=pkg.DIFFUSE)


def chain_rays(levels, n=4096, seed=0):
    """Rays along the chain in both directions (the -x ones fill the stack), slightly off axis."""
    rng = np.random.default_rng(seed)
    K = levels + 1
    org = np.zeros((n, 3), dtype=np.float32)
    d = np.zeros((n, 3), dtype=np.float32)
    back = rng.random(n) < 0.7
    org[:, 0] = np.where(back, K + 2.0, -2.0)
    org[:, 1:] = rng.uniform(-0.3, 0.3, size=(n, 2))
    d[:, 0] = np.where(back, -1.0, 1.0)
    d[:, 1:] = rng.uniform(-0.02, 0.02, size=(n, 2))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return org, d.astype(np.float32)


def handworked_hits(pkg):
    """A scene whose closest hits follow by hand, and rays with their answers (t, primitive id, unit normal), all exact in
    binary floating point.  The mesh is the six unit right triangles of tests/golden/bvh_handworked.npz (triangle k spans
    x in [a_k, a_k + 1], y in [0, 1] below the diagonal x - a_k + y <= 1, in the plane z = 1; a = 9, 0, 2, 12, 1, 10), over a
    plane z = -2 (normal +z) and a sphere of radius 0.5 at (4.5, 0.25, 0).
    Rays at the mesh have the direction (2^-10, 2^-10, +-1) -- not of unit length, which the triangle and plane tests do not
    ask for -- and start 2^-9 short of the point aimed at in x and y, so that they arrive there at t = 2, every product and sum
    on the way exact: with the Woop rows (0, 0, 1, 1), (1, 0, 0, -a), (0, 1, 0, 0) of these triangles (reference src/bvh.cu:214-227)
    t = (1 - o.z) / d.z, u = o.x - a + t d.x, v = o.y + t d.y (src/bvh.cuh:93-120).
      aimed at (a_k + 0.25, 0.25, 1): u = v = 0.25 -> triangle k at t = 2, normal cross(row x, row y) = (0, 0, 1), never flipped
        towards the ray (src/bvh.cuh:114), from below (z = -1) and from above (z = 3) alike;
      aimed at (a_k + 0.75, 0.75, 1): u + v = 1.5 -> not in the triangle: from below nothing else lies ahead (a miss), from above
        the plane at t = 5 (src/object.cuh:62-75).
    Rays exactly along +-z (unit length):  at x = 4.5, y = 0.25 the sphere, from below at t = 1 - 0.5, normal (0, 0, -1), from
    above at t = 3 - 0.5, normal (0, 0, 1) (src/object.cuh:27-44);  at x = 6.5, a gap of the mesh, nothing from below and the
    plane at t = 5 from above.  And the reference's treatment of axis-parallel rays at the mesh: with d.x = 0 its slab test
    forms x0 * inf - o.x * inf = NaN (src/bvh.cuh:79-80,136-137), which its integer min / max order above every number
    (src/bvh.cuh:18-27), so both boxes under the root are missed and the ray passes THROUGH triangle k: from below at
    (a_k + 0.25, 0.25) a miss, from above the plane at t = 5."""
    g = np.load(os.path.join(GOLDEN, "bvh_handworked.npz"))
    a = [9.0, 0.0, 2.0, 12.0, 1.0, 10.0]
    bvh = pkg.bvh_build(g["verts"])
    cam = pkg.camera_make((6, 0.5, 12), (6, 0.5, 0), 40.0, 1.5, 0.0, 12.0)
    tex, marg, cond = small_env(pkg)
    planes = [pkg.make_plane((0, 0, -2), (0, 0, 1), (0.5, 0.5, 0.5, 0.0), pkg.DIFFUSE)]
    spheres = [pkg.make_sphere((4.5, 0.25, 0.0), 0.5, (0.5, 0.5, 0.5, 0.0), pkg.MIRROR)]
    scene = pkg.ManualScene(cam, tex, marg, cond, planes=planes, spheres=spheres, bvh=bvh,
                            mesh_material=(0.6, 0.6, 0.6, 0.0), mesh_material_type=pkg.DIFFUSE)
    PLANE, SPHERE, MISS = 0x80000000, 0x40000000, 0xFFFFFFFF
    FLT_MAX = np.float32(3.402823466e38)
    e = 2.0 ** -10
    rays = []                                            # origin, direction, t, prim, normal
    for k, ak in enumerate(a):
        for z0, dz in ((-1.0, 1.0), (3.0, -1.0)):
            rays.append(((ak + 0.25 - 2 * e, 0.25 - 2 * e, z0), (e, e, dz), 2.0, k, (0, 0, 1)))
        rays.append(((ak + 0.75 - 2 * e, 0.75 - 2 * e, -1.0), (e, e, 1.0), FLT_MAX, MISS, (0, 0, 0)))
        rays.append(((ak + 0.75 - 2 * e, 0.75 - 2 * e, 3.0), (e, e, -1.0), 5.0, PLANE | 0, (0, 0, 1)))
        rays.append(((ak + 0.25, 0.25, -1.0), (0, 0, 1), FLT_MAX, MISS, (0, 0, 0)))           # axis-parallel: through the mesh
        rays.append(((ak + 0.25, 0.25, 3.0), (0, 0, -1), 5.0, PLANE | 0, (0, 0, 1)))
    rays.append(((4.5, 0.25, -1), (0, 0, 1), 0.5, SPHERE | 0, (0, 0, -1)))
    rays.append(((4.5, 0.25, 3), (0, 0, -1), 2.5, SPHERE | 0, (0, 0, 1)))
    rays.append(((6.5, 0.25, -1), (0, 0, 1), FLT_MAX, MISS, (0, 0, 0)))
    rays.append(((6.5, 0.25, 3), (0, 0, -1), 5.0, PLANE | 0, (0, 0, 1)))
    org = np.array([r[0] for r in rays], dtype=np.float32)
    dir_ = np.array([r[1] for r in rays], dtype=np.float32)
    t = np.array([r[2] for r in rays], dtype=np.float32)
    prim = np.array([r[3] for r in rays], dtype=np.uint32)
    nrm = np.array([r[4] for r in rays], dtype=np.float32)
    return scene, org, dir_, t, prim, nrm


def handworked_film(pkg, width=32, height=16):
    """A scene whose film follows by hand: a pinhole camera at (0, 1, 0) looking horizontally, a mirror plane y = 0 of albedo
    (0.5, 0.25, 0.75), a one-texel environment of radiance e = (2, 1, 0.5).  A camera ray that goes up leaves the scene: its
    pixel gets e.  One that goes down hits the mirror (reference src/render.cu:195-211: throughput x albedo, direction
    reflected upwards) and then leaves: its pixel gets albedo x e = (1, 0.25, 0.375).  A one-texel map returns its texel for
    every direction (the bilinear weights multiply differences that are zero), the values are dyadic, and every sample of a
    pixel contributes the same value divided by a power-of-two sample count, so the sums are exact whatever the jitter --
    except in the two pixel rows next to the horizon, where the jitter decides which side a sample falls on.
    Returns (scene, expected film with NaN in the two horizon rows)."""
    cam = pkg.camera_make((0, 1, 0), (0, 1, -1), 60.0, width / height, 0.0, 1.0)
    tex = np.zeros((1, 1, 4), dtype=np.float32)
    tex[0, 0] = [2.0, 1.0, 0.5, 1.0]
    marg, cond = pkg.env_build(tex)
    planes = [pkg.make_plane((0, 0, 0), (0, 1, 0), (0.5, 0.25, 0.75, 0.0), pkg.MIRROR)]
    scene = pkg.ManualScene(cam, tex, marg, cond, planes=planes)
    film = np.empty((height, width, 3), dtype=np.float32)
    film[:height // 2 - 1] = [2.0, 1.0, 0.5]                  # row 0 is the top of the image
    film[height // 2 - 1:height // 2 + 1] = np.nan
    film[height // 2 + 1:] = [1.0, 0.25, 0.375]
    return scene, film


def furnace_cases(pkg):
    """Scenes whose converged film is known from radiometry alone, not from any restatement of the reference: name ->
    (scene, film width, film height, expected mean colour).  Every pixel of each looks at one convex Lambertian, glossy or
    mirror surface that sees nothing but the environment, so a path is camera -> surface -> sky.
      "sky over a plane": a Lambertian plane y = 0 of albedo rho under a smooth, non-uniform sky L(w): every pixel converges to
        rho / pi x the integral of L(w) w.y over the upper hemisphere, here taken in float64 on a 2048 x 8192 midpoint grid.
      the others: a constant sky L (64 x 32 texels, so the light sampler and its pdf are in play) around a unit sphere or a
        convex 960-triangle mesh: Lambertian rho L; mirror rho L (reference src/render.cu:195-211); glossy (0.05 + 0.95 rho) L,
        the 5 % mirror branch leaving the throughput alone (src/render.cu:164-193); clear glass L; and a camera between two
        facing mirrors rho^8 L: the depth cut and the roulette.
    Scenes with interreflection are left out on purpose: the reference's roulette divides a throughput above 1 by itself
    without a matching survival probability (src/render.cu:78-86), so its multi-bounce films lose energy by design of the
    reference (a white furnace of the README's spheres over a floor comes out at 0.78-0.99 L), and physics pins nothing there.
    The reference's estimator (one-sample mix of cosine and light sampling, src/render.cu:129-162) is unbiased up to its light
    sampler, which draws texel-grid directions from two quantile tables of W and H entries (src/environment_map.cu:107-125,
    src/environment_map.cuh:39-49): that costs up to 0.4 % here, so the tests allow 1 %."""
    def sky(d):
        s = np.array([0.3, 0.8, 0.5]) / np.linalg.norm([0.3, 0.8, 0.5])
        return 0.6 + 0.4 * d[..., 0] + 1.5 * np.maximum(d @ s, 0.0) ** 8

    def dirs(theta, phi):
        T, P = np.meshgrid(theta, phi, indexing="ij")
        return np.stack([np.sin(T) * np.cos(P), np.cos(T), np.sin(T) * np.sin(P)], -1), T

    tint = np.array([1.0, 0.7, 0.4])
    rho = np.array([0.8, 0.5, 0.3])
    mat = (0.8, 0.5, 0.3, 0.0)
    cases = {}
    # texel (row r, column c) is looked up at theta = (r + 0.5) pi / H from +y and atan2(z, x) = ((c + 0.5) / W - 0.5) 2 pi
    # (reference src/environment_map.cuh:32-37)
    W, H = 256, 128
    d, _ = dirs((np.arange(H) + 0.5) * np.pi / H, ((np.arange(W) + 0.5) / W - 0.5) * 2 * np.pi)
    tex = np.zeros((H, W, 4), dtype=np.float32)
    tex[..., :3] = sky(d)[..., None] * tint
    marg, cond = pkg.env_build(tex)
    n = 2048
    d, T = dirs((np.arange(n) + 0.5) * (np.pi / 2) / n, (np.arange(4 * n) + 0.5) * 2 * np.pi / (4 * n))
    integral = (sky(d) * np.sin(T) * np.cos(T)).sum() * (np.pi / 2 / n) * (2 * np.pi / (4 * n))
    cam = pkg.camera_make((0, 1, 0), (0, 0, -1), 30.0, 2.0, 0.0, 1.0)
    planes = [pkg.make_plane((0, 0, 0), (0, 1, 0), mat, pkg.DIFFUSE)]
    cases["sky over a plane"] = (pkg.ManualScene(cam, tex, marg, cond, planes=planes), 32, 16, rho * tint * integral / np.pi)

    L = np.array([2.0, 1.0, 0.5])
    tex = np.zeros((32, 64, 4), dtype=np.float32)
    tex[..., :3] = L
    marg, cond = pkg.env_build(tex)
    cam = pkg.camera_make((0, 0, 4), (0, 0, 0), 10.0, 2.0, 0.0, 4.0)      # the whole film lies inside the silhouette
    for kind, typ, want in (("lambertian", pkg.DIFFUSE, rho * L), ("mirror", pkg.MIRROR, rho * L),
                            ("glossy", pkg.GLOSSY, (0.05 + 0.95 * rho) * L)):
        spheres = [pkg.make_sphere((0, 0, 0), 1.0, mat, typ)]
        cases[kind + " sphere"] = (pkg.ManualScene(cam, tex, marg, cond, spheres=spheres), 32, 16, want)
    # clear glass: every interaction multiplies the throughput by 1 (src/render.cu:213-242), the roulette's survival
    # probability stays 1, and a path cut at the depth limit still collects the sky (src/render.cu:62-73): L whatever the path
    spheres = [pkg.make_sphere((0, 0, 0), 1.0, (1.0, 1.0, 1.0, 1.5), pkg.GLASS)]
    cases["clear glass sphere"] = (pkg.ManualScene(cam, tex, marg, cond, spheres=spheres), 32, 16, L)
    # two facing mirrors, the camera between them: no ray gets out, every path is cut after its 8th hit (two depth increments
    # per hit against a limit of 16, src/render.cu:9,62-73) and collects the sky there, with rho^8 left of its throughput; the
    # roulette (from the second hit on, survival probability = largest channel <= 1, src/render.cu:78-86) does not change the
    # mean.  The camera looks 11 to 51 degrees upwards: a ray within epsilon of parallel to a plane passes it by the
    # reference's rule (|d.n| <= 0.001 is a miss, src/object.cuh:62-75), gets out and would add its larger throughput.
    hall = [pkg.make_plane((0, 0, 0), (0, 1, 0), (0.9, 0.8, 0.5, 0.0), pkg.MIRROR),
            pkg.make_plane((0, 2, 0), (0, -1, 0), (0.9, 0.8, 0.5, 0.0), pkg.MIRROR)]
    cam_hall = pkg.camera_make((0, 1, 0), (0, 1.6, -1), 40.0, 2.0, 0.0, 1.0)       # elevations 11 to 51 degrees
    cases["hall of mirrors"] = (pkg.ManualScene(cam_hall, tex, marg, cond, planes=hall), 32, 16,
                                np.array([0.9, 0.8, 0.5]) ** 8 * L)
    nu, nv = 32, 16

    def vertex(i, j):
        th, ph = np.pi * j / nv, 2 * np.pi * i / nu
        return (np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph))

    tris = []
    for j in range(nv):
        for i in range(nu):
            a, b, c, e = vertex(i, j), vertex(i + 1, j), vertex(i + 1, j + 1), vertex(i, j + 1)
            if j > 0:
                tris.append((a, b, c))                       # wound so that cross(v0 - v2, v1 - v2) points outwards
            if j < nv - 1:
                tris.append((a, c, e))
    bvh = pkg.bvh_build(np.array(tris, dtype=np.float32))
    for kind, typ, want in (("lambertian", pkg.DIFFUSE, rho * L), ("glossy", pkg.GLOSSY, (0.05 + 0.95 * rho) * L)):
        cases[kind + " mesh"] = (pkg.ManualScene(cam, tex, marg, cond, bvh=bvh, mesh_material=mat, mesh_material_type=typ),
                                 32, 16, want)
    return cases


def degenerate_mesh():
    """The low-poly bunny with 5 % of its triangles made degenerate: two equal vertices, three vertices on a line, all three
    equal.  The reference gives a triangle whose matrix is singular the identity as its transform (include/linear_math.cuh:
    112-113 via src/bvh.cu:214-227): a phantom unit triangle in the plane z = 0 at the origin, reachable only through the
    degenerate triangle's own (epsilon-padded) leaf box.  Returns (triangles, indices of the degenerate ones)."""
    tris = golden_mesh("bunny_lowpoly").copy()
    rng = np.random.default_rng(5)
    idx = np.sort(rng.choice(len(tris), len(tris) // 20, replace=False))
    tris[idx[::3], 1] = tris[idx[::3], 0]
    tris[idx[1::3], 2] = tris[idx[1::3], 0] + (tris[idx[1::3], 1] - tris[idx[1::3], 0]) * np.float32(0.5)
    tris[idx[2::3], 1] = tris[idx[2::3], 0]
    tris[idx[2::3], 2] = tris[idx[2::3], 0]
    return tris, idx


def fresnel_case(pkg, W=32, H=16):
    """A clear glass plane y = 0 (index 1.5) under a sky of colour A above the horizon, whose green grows linearly with the
    angle from the zenith, and, below it, a colour B whose red does; seen at 9 to 39 degrees of elevation by a pinhole camera.  A camera ray is
    reflected with probability R = r0 + (1 - r0) (1 - cos)^5, r0 = ((1 - eta) / (1 + eta))^2 = 0.04 (Schlick, reference
    src/material.cuh:34-39, chosen against one uniform number, src/render.cu:213-242) and then sees A at the mirrored
    angle (include/cutil_math.cuh:1289-1292); else it is refracted
    (src/material.cuh:26-32), goes on below the plane at the angle Snell's law gives, sin(theta_t) = sin(theta_i) / 1.5, and
    sees B there; the throughput stays 1.  So a pixel converges to R A(theta_i) + (1 - R) B(theta_t), with R from 0.05 at the bottom of
    the film to 0.44 at its top and theta_t from 31 to 41 degrees -- known from optics and the camera model
    (src/camera.cuh:39-53: u = (x + xi - 0.5) / (W - 1), rows counted from the bottom), averaged here over the pixel's
    jitter square in float64.  A ray that went straight through instead would see a red lower by 0.15 to 0.3.
    The environment's rows hold B at their centres, and the bilinear lookup reproduces a function that is linear in the
    angle exactly between row centres (src/environment_map.cuh:32-37).
    Returns (scene, expected film (H, W, 3), the tolerance unit |A - B| = 0.75)."""
    def above(theta_norm):                                   # angle from +y over pi, 0 .. 0.5: green grows towards the horizon
        return np.stack([np.ones_like(theta_norm), 0.5 + theta_norm, np.full_like(theta_norm, 0.25)], -1)

    def below(theta_norm):                                   # angle from +y over pi, 0.5 .. 1
        return np.stack([0.25 + 1.5 * (theta_norm - 0.5), np.full_like(theta_norm, 0.5), np.ones_like(theta_norm)], -1)

    tex = np.zeros((32, 64, 4), dtype=np.float32)
    tex[:16, :, :3] = above((np.arange(16) + 0.5) / 32)[:, None, :]       # rows 0..15: directions above the horizon
    tex[16:, :, :3] = below((np.arange(16, 32) + 0.5) / 32)[:, None, :]
    marg, cond = pkg.env_build(tex)
    cam = pkg.camera_make((0, 1, 0), (0, 0.55, -1), 30.0, W / H, 0.0, 1.0)
    planes = [pkg.make_plane((0, 0, 0), (0, 1, 0), (1.0, 1.0, 1.0, 1.5), pkg.GLASS)]
    scene = pkg.ManualScene(cam, tex, marg, cond, planes=planes)
    bl, right, up = (np.array(list(getattr(cam, f)), dtype=np.float64) for f in ("bottom_left", "right", "up"))
    g = (np.arange(8) + 0.5) / 8 - 0.5
    want = np.zeros((H, W, 3))
    for y in range(H):
        for x in range(W):
            u = (x + g[None, :]) / (W - 1)
            v = (y + g[:, None]) / (H - 1)
            t = bl + u[..., None] * right + v[..., None] * up
            cos_i = -t[..., 1] / np.linalg.norm(t, axis=-1)
            R = (0.04 + 0.96 * (1 - cos_i) ** 5)[..., None]
            cos_t = np.sqrt(1 - (1 - cos_i ** 2) / 1.5 ** 2)
            seen_above = above(np.arccos(cos_i) / np.pi)      # mirrored: the angle from the zenith is the angle of incidence
            seen_below = below(np.arccos(-cos_t) / np.pi)
            want[H - 1 - y, x] = (R * seen_above + (1 - R) * seen_below).mean(axis=(0, 1))
    return scene, want, 0.75


def shaded_sphere_case(pkg, W=32, H=16):
    """A Lambertian unit sphere under a smooth sky that is symmetric about no axis, seen from (1.5, 1, 3.5): the pixel whose
    centre ray meets the sphere at normal n converges to rho / pi x the integral of L(w) max(n.w, 0) over all directions,
    taken here in float64 on a 256 x 512 midpoint grid.  Unlike the plane of furnace_cases this holds the orientation of the
    environment map -- which texel a direction looks up (reference src/environment_map.cuh:32-37: column from atan2(z, x),
    row from acos(y)) and which direction a sampled texel stands for (src/environment_map.cuh:39-49) -- and the sphere's
    normal (src/object.cuh:39): the expected values vary from 0.54 to 0.84 over the film, a mirrored map or a flipped
    normal would move them by tens of percent.  Returns (scene, expected film (H, W, 3))."""
    def sky(d):
        s = np.array([0.3, 0.8, 0.5]) / np.linalg.norm([0.3, 0.8, 0.5])
        return 0.6 + 0.4 * d[..., 0] + 1.5 * np.maximum(d @ s, 0.0) ** 8

    def dirs(theta, phi):
        T, P = np.meshgrid(theta, phi, indexing="ij")
        return np.stack([np.sin(T) * np.cos(P), np.cos(T), np.sin(T) * np.sin(P)], -1), T

    tint = np.array([1.0, 0.7, 0.4])
    rho = np.array([0.8, 0.5, 0.3])
    EW, EH = 256, 128
    d, _ = dirs((np.arange(EH) + 0.5) * np.pi / EH, ((np.arange(EW) + 0.5) / EW - 0.5) * 2 * np.pi)
    tex = np.zeros((EH, EW, 4), dtype=np.float32)
    tex[..., :3] = sky(d)[..., None] * tint
    marg, cond = pkg.env_build(tex)
    cam = pkg.camera_make((1.5, 1.0, 3.5), (0, 0, 0), 10.0, W / H, 0.0, 4.0)      # the whole film lies on the sphere
    spheres = [pkg.make_sphere((0, 0, 0), 1.0, (0.8, 0.5, 0.3, 0.0), pkg.DIFFUSE)]
    scene = pkg.ManualScene(cam, tex, marg, cond, spheres=spheres)
    o = np.array(list(cam.origin), dtype=np.float64)
    bl, right, up = (np.array(list(getattr(cam, f)), dtype=np.float64) for f in ("bottom_left", "right", "up"))
    n = 256
    qd, T = dirs((np.arange(n) + 0.5) * np.pi / n, (np.arange(2 * n) + 0.5) * np.pi / n)
    qd = qd.reshape(-1, 3)
    Lq = sky(qd) * (np.sin(T) * (np.pi / n) ** 2).reshape(-1)
    want = np.zeros((H, W, 3))
    for y in range(H):
        for x in range(W):
            t = bl + x / (W - 1) * right + y / (H - 1) * up        # the pixel's centre ray (src/camera.cuh:39-53)
            dd = t / np.linalg.norm(t)
            b = o @ dd
            nrm = o + (-b - np.sqrt(b * b - (o @ o - 1.0))) * dd     # first intersection with the unit sphere = its normal
            want[H - 1 - y, x] = rho * tint * (Lq * np.maximum(qd @ nrm, 0.0)).sum() / np.pi
    return scene, want


def as_oracle_desc(scene):
    """The same scene bytes viewed through the oracle's (identically laid out) descriptor type."""
    import oracle_lib

    desc = scene.desc if hasattr(scene, "desc") else scene
    return C.cast(C.pointer(desc), C.POINTER(oracle_lib.SceneDesc)).contents


def secondary_rays(org, dir_, t, nrm, seed):
    """One bounce ray per hit: origin at the hit point, direction from a seeded cosine-ish lobe around the normal."""
    rng = np.random.default_rng(seed)
    hit = t < 3.0e38
    o = (org[hit] + dir_[hit] * t[hit, None]).astype(np.float32)
    r = rng.normal(size=o.shape).astype(np.float32)
    r /= np.linalg.norm(r, axis=1, keepdims=True)
    d = (nrm[hit] + r).astype(np.float32)
    ln = np.linalg.norm(d, axis=1, keepdims=True)
    d = np.where(ln > 1e-3, d / np.maximum(ln, 1e-12), nrm[hit]).astype(np.float32)
    return np.ascontiguousarray(o), np.ascontiguousarray(d)


def parity_cases(pkg, W=48, H=32):
    """name -> scene for the fixed ray-batch fixtures (tests/golden/rays_<name>.npz)."""
    return {
        "knot": knot_scene(pkg, W, H),
        "balls": spheres_scene(pkg, W, H),
        "bunny_lowpoly": mesh_scene(pkg, golden_mesh("bunny_lowpoly"), W, H, (1, 1.5, 4), (0, 1.0, 0),
                                    planes=[pkg.make_plane((0, 0, 0), (0, 1, 0), (0.3, 0.3, 0.3, 0), pkg.DIFFUSE)]),
        "suzanne": mesh_scene(pkg, golden_mesh("suzanne"), W, H, (0, -0.05, 4), (0, -0.05, 0)),
    }


def ray_batch(scene, W, H, seed):
    """Primary rays of the scene camera, one bounce ray per hit, and 256 axis-parallel / zero-component rays.

    Built with the CPU oracle (test infrastructure); returns (org, dir) float32 arrays.
    """
    import oracle_lib as O

    d = as_oracle_desc(scene)
    org, dir_ = O.primary_rays(d, W, H, sample=0, seed=seed)
    t, prim, nrm, _ = O.intersect(d, org, dir_, want_normals=True)
    o2, d2 = secondary_rays(org, dir_, t, nrm, seed + 1)
    rng = np.random.default_rng(seed + 2)
    k = 256
    o3 = rng.uniform(-1.5, 1.5, size=(k, 3)).astype(np.float32)
    d3 = np.zeros((k, 3), dtype=np.float32)
    ax = rng.integers(0, 3, size=k)
    d3[np.arange(k), ax] = rng.choice([-1.0, 1.0], size=k)
    two = rng.random(k) < 0.5
    ax2 = (ax + 1 + rng.integers(0, 2, size=k)) % 3
    d3[np.arange(k)[two], ax2[two]] = rng.choice([-1.0, 1.0], size=two.sum())
    d3 /= np.linalg.norm(d3, axis=1, keepdims=True)
    o3[::7, 0] = 0.0        # origins on a coordinate plane: 0 * inf in the slab set-up
    o3[::11, 1] = 0.0
    return (np.concatenate([org, o2, o3]).astype(np.float32), np.concatenate([dir_, d2, d3]).astype(np.float32))
