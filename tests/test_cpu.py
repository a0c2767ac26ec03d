"""CPU tests: the oracle against its pins and fixtures, the host-side scene assembly against the oracle, the C ABI
surface of both shared libraries, and the multi-rank film reduction (gloo, two processes)."""
import ctypes as C
import os
import re
import struct
import subprocess
import sys
import zlib

import numpy as np
import pytest

import oracle_lib as O
import scenes as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = S.GOLDEN


# ---- oracle pins -------------------------------------------------------------------------------------------------
def test_philox_published_vectors():
    """Philox4x32-10 known-answer vectors from the Random123 distribution (kat_vectors)."""
    assert [int(x) for x in O.philox([0, 0, 0, 0], [0, 0])] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert [int(x) for x in O.philox([0xffffffff] * 4, [0xffffffff] * 2)] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6,
                                                                              0x6d5451fd]
    assert [int(x) for x in O.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])] == [
        0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


# Sizes of the buffers the reference's own builder produces for its shipped meshes (node buffer bytes, triangle
# buffer bytes): 64 B per interior node; 16 B * (1 + 3*triangles + leaves).
REFERENCE_BUFFER_SIZES = {
    "bunny_lowpoly": (316, 11904, 18176),
    "suzanne": (968, 37312, 55824),
    "city": (14350, 491840, 811792),
    "bunny": (28576, 1049280, 1634000),
}


@pytest.mark.parametrize("name", list(REFERENCE_BUFFER_SIZES))
def test_oracle_bvh_structure(name):
    ntris, node_bytes, tri_bytes = REFERENCE_BUFFER_SIZES[name]
    tris = S.golden_mesh(name)
    assert tris.shape[0] == ntris
    b = O.bvh_build(tris)
    assert b["nodes"].nbytes == node_bytes
    assert b["num_tri_float4"] * 16 == tri_bytes
    assert b["num_leaves"] == b["nodes"].shape[0] + 1
    # every triangle appears exactly once, every leaf run is terminated
    ids = b["prim_ids"]
    assert sorted(ids[ids != O.PRIM_MISS].tolist()) == list(range(ntris))
    term = b["tris"][: b["num_tri_float4"], 0].view(np.uint32) == 0x80000000
    assert term.sum() == b["num_leaves"]
    # children: node indices in range, leaf references inside the triangle buffer
    ch = b["nodes"][:, 12:14].view(np.int32)
    assert ((ch >= 0) & (ch < b["nodes"].shape[0]) | (ch < 0) & (~ch < b["num_tri_float4"])).all()
    # the root's two boxes enclose the mesh
    n0 = b["nodes"][0]
    lo = np.minimum([n0[0], n0[2], n0[8]], [n0[4], n0[6], n0[10]])
    hi = np.maximum([n0[1], n0[3], n0[9]], [n0[5], n0[7], n0[11]])
    assert (lo <= tris.reshape(-1, 3).min(axis=0)).all() and (hi >= tris.reshape(-1, 3).max(axis=0)).all()


@pytest.mark.parametrize("name", ["knot", "balls", "bunny_lowpoly", "suzanne"])
def test_oracle_matches_golden_rays(pkg, name):
    g = np.load(os.path.join(GOLDEN, "rays_%s.npz" % name))
    scene = S.parity_cases(pkg)[name]
    t, prim, _, _ = O.intersect(S.as_oracle_desc(scene), g["org"], g["dir"])
    assert np.array_equal(prim, g["prim"])
    assert np.array_equal(t, g["t"])


def test_oracle_woop_hits_agree_with_moller_trumbore(pkg):
    """Independent check of the traversal + Woop test: brute-force float64 Moller-Trumbore over every triangle."""
    tris = S.golden_mesh("bunny_lowpoly")
    scene = S.parity_cases(pkg)["bunny_lowpoly"]
    d = S.as_oracle_desc(scene)
    org, dir_ = O.primary_rays(d, 48, 32, sample=0, seed=11)
    # only the mesh: compare against rays' nearest triangle
    t, prim, _, _ = O.intersect(d, org, dir_)
    v0, v1, v2 = (tris[:, k].astype(np.float64) for k in range(3))
    e1, e2 = v1 - v0, v2 - v0
    checked = 0
    for i in range(org.shape[0]):
        if prim[i] >= O.PRIM_SPHERE:
            continue
        o, dd = org[i].astype(np.float64), dir_[i].astype(np.float64)
        p = np.cross(dd, e2)
        det = (e1 * p).sum(1)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / det
            s = o - v0
            u = (s * p).sum(1) * inv
            q = np.cross(s, e1)
            v = (q * dd).sum(1) * inv
            tt = (e2 * q).sum(1) * inv
        ok = (np.abs(det) > 1e-12) & (u >= -1e-9) & (v >= -1e-9) & (u + v <= 1 + 1e-9) & (tt > 1e-3)
        assert ok.any()
        best = np.where(ok, tt, np.inf).min()
        assert abs(best - t[i]) <= 1e-4 * max(1.0, best)
        checked += 1
    assert checked > 100


def test_oracle_render_energy_and_determinism(pkg):
    scene = S.spheres_scene(pkg, 48, 32)
    d = S.as_oracle_desc(scene)
    a, sa = O.render(d, 48, 32, 0, 4, 4, seed=2, nthreads=1)
    b, sb = O.render(d, 48, 32, 0, 4, 4, seed=2, nthreads=4)
    assert np.array_equal(a, b) and sa.rays == sb.rays
    assert np.isfinite(a).all() and a.min() >= 0 and a.mean() > 0.05
    # sample partition: two halves sum to the whole
    h0, _ = O.render(d, 48, 32, 0, 2, 4, seed=2)
    h1, _ = O.render(d, 48, 32, 2, 2, 4, seed=2)
    assert np.allclose(h0 + h1, a, rtol=1e-6, atol=1e-7)


# ---- host-side scene assembly against the oracle ---------------------------------------------------------------------
@pytest.mark.parametrize("name", ["bunny_lowpoly", "suzanne", "city", "bunny"])
@pytest.mark.parametrize("threads", [1, 4])
def test_host_bvh_equals_oracle(pkg, name, threads):
    tris = S.golden_mesh(name)
    a = pkg.bvh_build(tris, threads=threads)
    b = O.bvh_build(tris)
    for k in ("nodes", "tris", "prim_ids"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
    assert a["max_depth"] == b["max_depth"] and a["num_leaves"] == b["num_leaves"]


def test_bvh_builders_reproduce_the_handworked_tree(pkg):
    """tests/golden/bvh_handworked.npz holds the node and triangle buffers of a six-triangle mesh worked out by hand from the
    reference's text (bucket choice incl. a tie, std::partition's element order, ordering of two-triangle leaves, right-first
    flatten, ~index leaf references, terminators, the zero pad, Woop rows; derivation in tools/make_golden_bvh_handworked.py,
    which calls no builder).  The product's builder, with one thread and with four, and the oracle's reproduce it byte for
    byte."""
    g = np.load(os.path.join(S.GOLDEN, "bvh_handworked.npz"))
    for what, b in (("oracle", O.bvh_build(g["verts"])), ("host, 1 thread", pkg.bvh_build(g["verts"], threads=1)),
                    ("host, 4 threads", pkg.bvh_build(g["verts"], threads=4))):
        assert b["nodes"].shape == (3, 16) and b["num_tri_float4"] == 23, what
        assert b["nodes"].tobytes() == g["nodes"].tobytes(), what
        assert b["tris"][:23].tobytes() == g["tris"].tobytes(), what
        assert not b["tris"][23:].any(), what                       # the two float4 the leaf loop may read past the end
        assert np.array_equal(b["prim_ids"][g["prim_index"]], g["prim_id"]), what
        assert (b["num_tris"], b["num_leaves"]) == (6, 4), what


def test_oracle_reproduces_handworked_hits(pkg):
    """Closest hits that follow by hand (tests/scenes.py::handworked_hits: unit triangles in z = 1, a plane, a sphere, rays
    along +-z; every t, id and normal exact): the oracle returns exactly them."""
    scene, org, dir_, t, prim, nrm = S.handworked_hits(pkg)
    ot, oprim, onrm, _ = O.intersect(S.as_oracle_desc(scene), org, dir_, want_normals=True)
    assert np.array_equal(oprim, prim)
    assert np.array_equal(ot, t)
    hit = prim != 0xFFFFFFFF
    assert np.array_equal(onrm[hit], nrm[hit])


def test_oracle_reproduces_handworked_film(pkg):
    """A film that follows by hand (tests/scenes.py::handworked_film: pinhole camera, mirror plane, one-texel environment):
    sky pixels hold the environment's radiance, floor pixels albedo times it, exactly, at 1, 4 and 64 samples per pixel."""
    scene, want = S.handworked_film(pkg)
    known = ~np.isnan(want)
    H, W = want.shape[:2]
    for spp in (1, 4, 64):
        film, st = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=spp)
        assert np.array_equal(film[known], want[known]), spp
        assert st.paths == W * H * spp and st.paths < st.rays <= 2 * st.paths


def test_oracle_converges_to_radiometry(pkg):
    """Films known from radiometry alone (tests/scenes.py::furnace_cases: a Lambertian plane under a non-uniform sky against
    the float64 integral of L cos / pi; Lambertian, mirror and glossy convex bodies in a constant sky): the oracle's mean over
    the film, 524,288 paths, is within 1 % (the statistical error is below 0.1 %; the rest is the reference's table-driven
    light sampler).  A check of the estimator -- pdfs, the 50/50 mix, the 5 % glossy branch, normals of spheres and of mesh
    triangles -- that is not one restatement agreeing with another."""
    for name, (scene, W, H, want) in S.furnace_cases(pkg).items():
        film, st = O.render(S.as_oracle_desc(scene), W, H, 0, 1024, 1024, seed=3, nthreads=8)
        got = film.reshape(-1, 3).mean(axis=0, dtype=np.float64)
        assert st.rays == 2 * st.paths or name in ("clear glass sphere", "hall of mirrors"), name   # camera -> surface -> sky
        assert np.abs(got / want - 1).max() < 0.01, (name, got, want)


def test_oracle_follows_the_fresnel_curve(pkg):
    """tests/scenes.py::fresnel_case: a glass plane between two sky colours; the row means of the oracle's film (32,768
    samples per row) follow Schlick's reflectance and Snell's refraction angle over the camera's elevations within 1 % of
    the colour difference, and R does span 0.05 to 0.44 over the film."""
    scene, want, span = S.fresnel_case(pkg)
    H, W = want.shape[:2]
    film, st = O.render(S.as_oracle_desc(scene), W, H, 0, 1024, 1024, seed=2, nthreads=8)
    assert st.rays == 2 * st.paths
    assert np.abs(film.mean(axis=1, dtype=np.float64) - want.mean(axis=1)).max() < 0.01 * span
    R = (want[..., 2] - 1.0) / (0.25 - 1.0)              # the blue channel is A's with R, B's with 1 - R
    assert R.min() < 0.06 and R.max() > 0.40


def test_oracle_shades_a_sphere_as_radiometry_says(pkg):
    """tests/scenes.py::shaded_sphere_case: 4 x 4-pixel block means of the oracle's film (32,768 paths each) are within 1.5 %
    of rho / pi x the cosine-weighted integral of the sky around each pixel's normal -- the orientation of the environment
    map and the sphere's normals, against a float64 quadrature."""
    scene, want = S.shaded_sphere_case(pkg)
    H, W = want.shape[:2]
    film, st = O.render(S.as_oracle_desc(scene), W, H, 0, 2048, 2048, seed=4, nthreads=8)
    assert st.rays == 2 * st.paths
    blocks = lambda a: a.reshape(H // 4, 4, W // 4, 4, 3).mean(axis=(1, 3), dtype=np.float64)
    assert np.abs(blocks(film) / blocks(want) - 1).max() < 0.015
    assert want[..., 0].max() / want[..., 0].min() > 1.5            # the film is far from flat


def test_host_bvh_parallel_build_equals_serial(pkg):
    """A mesh large enough for the parallel path: ranges of >= 65,536 triangles are split with all threads (parallel
    bounds, histogram and an order-preserving parallel partition), the subtrees below are built and flattened as tasks."""
    tris = S.knot_mesh(pkg)
    rng = np.random.default_rng(0)
    big = np.concatenate([tris + rng.uniform(-3, 3, size=(1, 1, 3)).astype(np.float32) for _ in range(12)])
    a = pkg.bvh_build(big, threads=1)
    for threads in (2, 3, 8):
        b = pkg.bvh_build(big, threads=threads)
        for k in ("nodes", "tris", "prim_ids"):
            assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), (threads, k)
        assert a["max_depth"] == b["max_depth"] and a["num_leaves"] == b["num_leaves"]


@pytest.mark.parametrize("case", ["outliers", "coincident", "signed_zero"])
def test_host_bvh_parallel_top_edge_cases(pkg, case):
    """Shapes of the tree top the parallel path has to get right, against the oracle's sequential builder: a split
    that leaves a single triangle on one side (a leaf directly below a top node), a large range of coincident
    centroids (median fallback at the top) and boxes that touch +0 and -0 (min/max ties must pick as a fold does)."""
    tris = S.knot_mesh(pkg)
    rng = np.random.default_rng(1)
    big = np.concatenate([tris + rng.uniform(-2, 2, size=(1, 1, 3)).astype(np.float32) for _ in range(8)])
    if case == "outliers":
        far = np.array([[[900, 0, 0], [901, 0, 0], [900, 1, 0]], [[-900, 5, 0], [-901, 5, 0], [-900, 6, 0]]], np.float32)
        big = np.concatenate([far[:1], big, far[1:]])
    elif case == "coincident":
        same = np.repeat(np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], np.float32), 70000, axis=0)
        big = np.concatenate([same, big[:5000]])
    else:
        big = big.copy()
        big[::7, :, 0] = np.where(big[::7, :, 0] > 0, np.float32(0.0), np.float32(-0.0))
        big[3::11, :, 1] = np.float32(-0.0)
    a = pkg.bvh_build(big, threads=4)
    b = O.bvh_build(big)
    for k in ("nodes", "tris", "prim_ids"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
    assert a["max_depth"] == b["max_depth"] and a["num_leaves"] == b["num_leaves"]


def test_host_bvh_rejects_tiny_meshes(pkg):
    with pytest.raises(pkg.PtError):
        pkg.bvh_build(np.zeros((2, 3, 3), np.float32))


def test_host_bvh_degenerate_centroids(pkg):
    """Identical triangles: the reference divides 0/0 here; both builders fall back to a median split."""
    tri = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], np.float32)
    tris = np.repeat(tri, 9, axis=0)
    a = pkg.bvh_build(tris)
    b = O.bvh_build(tris)
    assert np.array_equal(a["nodes"].view(np.uint32), b["nodes"].view(np.uint32))
    assert np.array_equal(a["tris"].view(np.uint32), b["tris"].view(np.uint32))


def test_host_bvh_with_degenerate_triangles(pkg):
    """Triangles without area (tests/scenes.py::degenerate_mesh): the builder gives them the identity transform like the
    reference does, never a NaN or an infinity, byte for byte as the oracle."""
    tris, idx = S.degenerate_mesh()
    b = O.bvh_build(tris)
    for threads in (1, 4):
        a = pkg.bvh_build(tris, threads=threads)
        for k in ("nodes", "tris", "prim_ids"):
            assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), (k, threads)
    assert np.isfinite(b["tris"]).all()
    rows = b["tris"][:b["num_tri_float4"]]
    first = np.flatnonzero(np.isin(b["prim_ids"][:len(rows)], idx))
    identity = np.array([[0, 0, 1, 0], [1, 0, 0, 0], [0, 1, 0, 0]], np.float32)
    singular = [i for i in first if np.array_equal(rows[i:i + 3], identity)]
    assert len(singular) >= len(idx) * 2 // 3            # two equal vertices and three equal vertices are exactly singular


def test_host_env_equals_oracle(pkg):
    tex, marg, cond = S.small_env(pkg, 64, 32)
    tex2 = tex.copy()
    tex2[..., 3] = 1.0
    m2, c2 = O.env_build(tex2)
    assert np.array_equal(tex.view(np.uint32), tex2.view(np.uint32))
    assert np.array_equal(marg, m2) and np.array_equal(cond, c2)
    # the bottom row's pdf is zero (sin(0) == 0), the pdf integrates to about one over the sphere
    assert (tex[-1, :, 3] == 0).all()
    H, W = tex.shape[:2]
    theta = (np.arange(H) + 0.5) * np.pi / H
    integral = (tex[..., 3] * np.sin(theta)[:, None]).sum() * (np.pi / H) * (2 * np.pi / W)
    assert 0.8 < integral < 1.2


def test_environment_tables_worked_out_by_hand(pkg):
    """The importance-sampling tables of a 4 x 2 map, derived on paper from the reference's text (src/environment_map.cu:55-125),
    not from either builder.  Grey texels of luminance L(x, y), distribution row y = 0 at the BOTTOM of the image
    (data index (H - y - 1) W + x):  L = (1, 3), (3, 1), (1, 2), (6, 2) for x = 0..3, y = 0, 1.
    Row weights sin((y + 0.5) pi / 2) are sin(pi / 4) for both rows and cancel.
      conditional CDFs per column: (1/4, 1), (3/4, 1), (1/3, 1), (3/4, 1); column sums 4, 4, 3, 8 of 19
      marginal CDF: 4/19, 8/19, 11/19, 1
      marginal table, entry x = first i with cdf[i] >= (x + 1) / 4, i never going back: 1, 2, 3, 3
      conditional tables, entry y = first i with cdf[i] >= (y + 1) / 2: (1, 1), (0, 1), (1, 1), (0, 1)
      pdf (written into alpha): row y = 0 has sin(0) = 0 -> 0; row y = 1 has sin(pi / 2) = 1 ->
        W H pdf_x pdf_y / (2 pi^2) with pdf_x = (4, 4, 3, 8) / 19 and pdf_y = cdf[1] - cdf[0] = 3/4, 1/4, 2/3, 1/4.
    No comparison above is near a tie, so the integer tables must come out exactly; the pdfs within float rounding."""
    L = np.array([[1, 3], [3, 1], [1, 2], [6, 2]], np.float32)               # [x][y]
    W, H = 4, 2
    want_marg = [1, 2, 3, 3]
    want_cond = [1, 1, 0, 1, 1, 1, 0, 1]                                     # [x * H + y]
    pdf_x = np.array([4, 4, 3, 8]) / 19.0
    pdf_y1 = np.array([3 / 4, 1 / 4, 2 / 3, 1 / 4])
    want_alpha = np.zeros((H, W))
    want_alpha[0] = W * H * pdf_x * pdf_y1 / (2 * np.pi ** 2)                 # y = 1 is image row 0
    for what, build in (("host", pkg.env_build), ("oracle", O.env_build)):
        tex = np.zeros((H, W, 4), np.float32)
        for x in range(W):
            for y in range(H):
                tex[H - 1 - y, x, :3] = L[x, y]
        tex[..., 3] = 7.0                                                    # overwritten by the build
        marg, cond = build(tex)
        assert list(marg) == want_marg, what
        assert list(cond) == want_cond, what
        assert np.array_equal(tex[1, :, 3], np.zeros(W, np.float32)), what
        assert np.allclose(tex[0, :, 3], want_alpha[0], rtol=2e-6, atol=0), (what, tex[0, :, 3], want_alpha[0])
        assert np.array_equal(tex[..., :3].max(axis=2), L.T[::-1]), what     # the colours are left alone


def test_host_camera_equals_oracle(pkg):
    for args in [((0, 1, 5), (0, 0.5, 0), 35.0, 16 / 9, 0.1, 4.75), ((-1.1, 0.2, -0.8), (0, 0, 0), 38.0, 1.5, 0.01, 1.15)]:
        assert bytes(pkg.camera_make(*args)) == bytes(O.camera_make(*args))
    # and a frame that follows on paper from the reference's text (src/camera.cuh:21-37): at the origin looking down -z, 90
    # degrees, aspect 2, focus distance 3, aperture 0.5: w = (0, 0, 1), u = (1, 0, 0), v = (0, 1, 0); the image plane at the
    # focus distance is 2 tan(45 deg) 3 = 6 high and 12 wide: right = (12, 0, 0), up = (0, 6, 0), bottom left (-6, -3, -3)
    for make in (pkg.camera_make, O.camera_make):
        cam = make((0, 0, 0), (0, 0, -1), 90.0, 2.0, 0.5, 3.0)
        got = [list(getattr(cam, f)) for f in ("origin", "bottom_left", "right", "up", "u", "v", "w")]
        want = [[0, 0, 0], [-6, -3, -3], [12, 0, 0], [0, 6, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]]
        assert np.allclose(got, want, rtol=0, atol=2e-6)
        assert cam.lens_radius == 0.25


def test_thin_lens_rays_meet_on_the_focal_plane(pkg):
    """What a thin lens does, not how the reference writes it (src/camera.cuh:39-53): with the aperture open, the ray of a
    film position starts somewhere on the lens disc and passes through the same point of the focal plane as the pinhole ray
    of that position.  The oracle's camera rays with aperture 0.3 and with aperture 0 (same seed, so the same jitter) meet
    there; the lens origins fill a disc of radius 0.15 in the lens plane evenly (mean 0, mean r^2 = R^2 / 2)."""
    W, H = 48, 32
    pos, look, focus = np.array([0.5, 1.0, 4.0]), np.array([0.0, 0.5, 0.0]), 3.0
    tex, marg, cond = S.small_env(pkg)
    rays = {}
    for aperture in (0.0, 0.3):
        cam = pkg.camera_make(tuple(pos), tuple(look), 35.0, W / H, aperture, focus)
        scene = pkg.ManualScene(cam, tex, marg, cond)
        rays[aperture] = tuple(a.astype(np.float64) for a in O.primary_rays(S.as_oracle_desc(scene), W, H, sample=3, seed=5))
    axis = (look - pos) / np.linalg.norm(look - pos)

    def on_focal_plane(org, d):
        t = (focus - (org - pos) @ axis) / (d @ axis)
        return org + d * t[:, None]

    (o0, d0), (o1, d1) = rays[0.0], rays[0.3]
    assert np.abs(o0 - pos).max() < 1e-6                                  # the pinhole
    assert np.abs(on_focal_plane(o0, d0) - on_focal_plane(o1, d1)).max() < 2e-5
    off = o1 - pos
    assert np.abs(off @ axis).max() < 1e-6                                # origins lie in the lens plane
    r2 = (off ** 2).sum(axis=1)
    assert r2.max() <= 0.15 ** 2 * (1 + 1e-5)
    assert abs(r2.mean() / (0.15 ** 2 / 2) - 1) < 0.06 and np.abs(off.mean(axis=0)).max() < 0.01
    assert np.abs(np.linalg.norm(d1, axis=1) - 1).max() < 1e-6


def test_hdr_roundtrip_and_rgbe_decode(pkg, tmp_path):
    rng = np.random.default_rng(1)
    img = (rng.random((16, 40, 3)) ** 4 * 100).astype(np.float32)
    img[3, :, :] = 0.0                     # an all-zero scanline: long runs
    img[5, 10:30, :] = 7.5                 # a run inside a scanline
    path = str(tmp_path / "t.hdr")
    pkg.hdr_write(path, img)
    back = pkg.hdr_load(path)
    assert back.shape == (16, 40, 4) and (back[..., 3] == 1).all()
    # RGBE keeps 8 mantissa bits of the largest channel
    mx = img.max(axis=2, keepdims=True)
    assert np.all(np.abs(back[..., :3] - img) <= mx / 128 + 1e-30)
    # decoding agrees with the oracle's restatement of stb_image's conversion
    raw = open(path, "rb").read()
    assert raw.startswith(b"#?RADIANCE\n")
    rgbe = np.array([[128, 64, 32, 130], [0, 0, 0, 0], [255, 1, 2, 120], [7, 7, 7, 1]], np.uint8)
    out = np.zeros((4, 4), np.float32)
    O.lib().orc_rgbe_to_float(O.ptr(rgbe), 4, O.ptr(out))
    assert out[0].tolist() == [2.0, 1.0, 0.5, 1.0] and out[1].tolist() == [0, 0, 0, 1]
    # a flat (non run-length) file decodes too
    flat = str(tmp_path / "flat.hdr")
    with open(flat, "wb") as fh:
        fh.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 2 +X 8\n")
        px = np.tile(np.array([[128, 64, 32, 130]], np.uint8), (16, 1))
        fh.write(px.tobytes())
    f = pkg.hdr_load(flat)
    assert f.shape == (2, 8, 4) and np.allclose(f[..., :3], [2.0, 1.0, 0.5])


def test_tonemap_and_png(pkg, tmp_path):
    rng = np.random.default_rng(2)
    film = (rng.random((9, 13, 3)) * 3).astype(np.float32)
    a = pkg.tonemap16(film)
    b = O.tonemap16(film)
    assert np.array_equal(a, b)
    # known answers of 65535 (1 - exp(-exposure x))^(1 / gamma), truncated (reference src/image.cpp:12-16): black stays 0, a bright
    # value saturates, and x = ln 2 / exposure gives 65535 * 0.5^(1 / 2.2) = 47823.6
    known = np.array([[[0.0, 100.0, np.log(2.0) / 2.0]]], np.float32)
    for tm in (pkg.tonemap16, O.tonemap16):
        v = tm(known).reshape(3).astype(int)
        assert v[0] == 0 and v[1] == 65535 and abs(v[2] - 47823) <= 1
    path = str(tmp_path / "o.png")
    pkg.png16_write(path, a)
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, {}
    while pos < len(raw):
        n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
        data = raw[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(typ + data) == crc
        chunks[typ] = data
        pos += 12 + n
    w, h, depth, ctype = struct.unpack(">IIBB", chunks[b"IHDR"][:10])
    assert (w, h, depth, ctype) == (13, 9, 16, 2)
    rows = zlib.decompress(chunks[b"IDAT"])
    stride = 1 + 13 * 6
    px = np.frombuffer(b"".join(rows[r * stride + 1:(r + 1) * stride] for r in range(9)), dtype=">u2").reshape(9, 13, 3)
    assert np.array_equal(px.astype(np.uint16), a)


def test_command_line_usage(pkg):
    """Wrong argument count: the usage line and exit status 0, as the reference's main does (src/main.cu:16-19)."""
    import subprocess
    r = subprocess.run([os.path.join(pkg.PKG_DIR, "pathtracer")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0
    assert r.stdout.strip() == "usage: ./pathtracer scene_file x_res y_res samples_per_pixel image_file"


def test_scene_file_loading(pkg, tmp_path):
    """A .scene file in the reference's format: blocks, materials, and asset paths relative to the scene file."""
    tris = S.golden_mesh("bunny_lowpoly")
    S.write_obj(str(tmp_path / "m.obj"), tris)
    env = np.ones((8, 16, 3), np.float32)
    env[2, 3] = 50.0
    pkg.hdr_write(str(tmp_path / "e.hdr"), env)
    text = """CAMERA
position 1 1 4
look_at -1 0.5 0
fov 35
aperture 0.1
focus_dist 4

ENVIRONMENT
file_name e.hdr

PLANE
material_type LAMBERTIAN
albedo 0.08 0.08 0.12
position 0 0 0
normal 0 2 0

SPHERE
material_type GLASS
albedo 1 1 1
refraction_index 1.5
position -3 1.5 -0.5
radius 1.5

OBJ
material_type GLOSSY
albedo 0.6 0.3 0.1
specular 1 1 1
glossyness 0.05
file_name m.obj
"""
    (tmp_path / "s.scene").write_text(text)
    sc = pkg.HostScene(str(tmp_path / "s.scene"), 64, 48)
    d = sc.desc
    assert d.num_planes == 1 and d.num_spheres == 1 and d.num_tris == 316
    assert d.mesh_material_type == pkg.GLOSSY and abs(d.mesh_material[3] - 0.05) < 1e-7
    pl = C.cast(d.planes, C.POINTER(pkg.Plane))[0]
    assert list(pl.normal) == [0.0, 1.0, 0.0]
    sp = C.cast(d.spheres, C.POINTER(pkg.Sphere))[0]
    assert sp.material_type == pkg.GLASS and sp.material[3] == 1.5 and sp.radius == 1.5
    assert (d.env_width, d.env_height) == (16, 8)
    assert bytes(d.camera) == bytes(O.camera_make((1, 1, 4), (-1, 0.5, 0), 35.0, 64 / 48, 0.1, 4.0))
    # the mesh buffers are the ones the builder gives for the same triangles
    b = O.bvh_build(tris)
    nodes = np.ctypeslib.as_array(C.cast(d.bvh_nodes, C.POINTER(C.c_float)), shape=(d.num_nodes, 16))
    assert np.array_equal(nodes.view(np.uint32), b["nodes"].view(np.uint32))
    # the oracle can render it
    film, st = O.render(S.as_oracle_desc(sc), 64, 48, 0, 1, 1)
    assert st.paths == 64 * 48 and np.isfinite(film).all()
    sc.close()
    with pytest.raises(pkg.PtError):
        pkg.HostScene(str(tmp_path / "missing.scene"), 64, 48)


def test_obj_loader_superset(pkg, tmp_path):
    (tmp_path / "a.obj").write_text("# comment\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\n\nf 1/1/1 2/2/1 3/3/1 4/4/1\nf -4 -3 -2\n")
    t = pkg.obj_load(str(tmp_path / "a.obj"))
    assert t.shape == (3, 3, 3)
    assert t[0].tolist() == [[0, 0, 0], [1, 0, 0], [1, 1, 0]]
    assert t[1].tolist() == [[0, 0, 0], [1, 1, 0], [0, 1, 0]]
    assert t[2].tolist() == [[0, 0, 0], [1, 0, 0], [1, 1, 0]]


def test_generated_assets(pkg):
    knot = S.knot_mesh(pkg)
    assert knot.shape == (9216, 3, 3)
    assert abs(float(knot[..., 1].min()) + 0.283) < 1e-6
    b = pkg.bvh_build(knot)
    assert b["max_depth"] <= 63


# ---- C ABI surface -----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("header,lib", [("pt_api.h", "libpt_hip.so"), ("pt_host.h", "libpt_host.so")])
def test_c_abi_exports_every_declared_symbol(header, lib):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(pth?_[a-z0-9_]+)\s*\(", text))
    assert len(names) >= 10
    path = os.path.join(ROOT, "cuda-path-tracer_amd", lib)
    assert os.path.exists(path), "%s is not built" % lib
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    assert names <= exported, "missing: %s" % sorted(names - exported)


def test_hip_library_loads_and_reports_missing_gpu(pkg):
    """Without a GPU the device library still loads; creating a context fails with a message instead of falling back."""
    import torch

    L = pkg.hip_lib()
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert L.pt_ctx_create(0, C.byref(h)) != 0
    assert b"HIP" in L.pt_last_error(None) or b"device" in L.pt_last_error(None)
    with pytest.raises(pkg.PtError):
        pkg.Context(0)


# ---- multi-rank: samples-per-pixel partition + film reduction ---------------------------------------------------------
def test_spp_partition(pkg):
    for spp, ws in [(512, 8), (7, 3), (2, 4), (2048, 8)]:
        parts = [pkg.spp_partition(spp, ws, r) for r in range(ws)]
        assert parts[0][0] == 0 and sum(c for _, c in parts) == spp
        for (b0, c0), (b1, _) in zip(parts, parts[1:]):
            assert b0 + c0 == b1


def test_two_rank_film_reduce_gloo(tmp_path):
    """world_size 2 over gloo: each rank renders its share of the samples (CPU oracle standing in for the device),
    the films are summed onto rank 0 with one reduce, and the result equals the single-rank film."""
    script = os.path.join(ROOT, "tests", "dist_worker.py")
    out = str(tmp_path / "film.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + os.getpid() % 2000))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", env["MASTER_PORT"], script, out]
    subprocess.check_call(cmd, env=env, cwd=ROOT, timeout=300)
    got = np.load(out)
    import __graft_entry__ as G

    pkg = G.load_package()
    scene = S.spheres_scene(pkg, 48, 32)
    want, _ = O.render(S.as_oracle_desc(scene), 48, 32, 0, 6, 6, seed=4)
    assert np.allclose(got, want, rtol=1e-6, atol=1e-7)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (no WORLD_SIZE) starts two rank processes itself, before it has
    touched a GPU, and reports their failure with a non-zero exit code instead of hanging.  On a machine without a GPU both
    ranks come up, rendezvous over gloo on 127.0.0.1 and then stop at the device ("No HIP GPUs"); on a GPU box the
    rehearsal form of the same command renders (tests/test_gpu_parity.py)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "rank exit codes [1, 1]" in r.stderr
    assert "[rank0]" in r.stderr and "[rank1]" in r.stderr          # both ranks ran, each under its own RANK
    assert "No HIP GPUs" in r.stderr
    assert r.stdout.strip() == ""


def test_oracle_film_matches_frozen_fixture(pkg):
    """The oracle's film of the knot scene equals the frozen one (tests/golden/film_knot.npz, written by
    tools/make_golden_film.py): scene construction, shading restatement and random streams have not drifted.
    libm's sin/cos/atan2/acos/pow are the only platform dependence, hence the 1e-6 relative slack."""
    g = np.load(os.path.join(S.GOLDEN, "film_knot.npz"))
    W, H, spp, seed = int(g["width"]), int(g["height"]), int(g["spp"]), int(g["seed"])
    scene = S.knot_scene(pkg, W, H)
    film, st = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=seed)
    assert st.paths == W * H * spp == int(g["paths"])
    assert st.rays == int(g["rays"])
    assert np.allclose(film, g["film"], rtol=1e-6, atol=1e-7)


def test_pfm_writer_keeps_the_exact_floats(pkg, tmp_path):
    rng = np.random.default_rng(5)
    film = rng.random((7, 5, 3), dtype=np.float32) * 40
    path = str(tmp_path / "f.pfm")
    pkg.pfm_write(path, film)
    raw = open(path, "rb").read()
    header = b"PF\n5 7\n-1.0\n"
    assert raw.startswith(header)
    back = np.frombuffer(raw[len(header):], dtype="<f4").reshape(7, 5, 3)[::-1]
    assert np.array_equal(back, film)


def test_scene_with_two_obj_blocks(pkg, tmp_path):
    """Several OBJ blocks (the reference stops at one, src/scene.cu:187-188): one BVH over all triangles in block order,
    a material per mesh; a one-OBJ scene still describes itself exactly as before."""
    scene = pkg.HostScene(S.scene_file(pkg, "two_meshes"), 64, 48)
    d = scene.desc
    a, b = S.two_meshes_triangles()
    assert d.num_meshes == 2 and d.num_tris == a.shape[0] + b.shape[0] == 316 + 968
    meshes = C.cast(d.meshes, C.POINTER(pkg.Mesh * 2)).contents
    assert (meshes[0].first_face, meshes[1].first_face) == (0, 316)
    assert (meshes[0].material_type, meshes[1].material_type) == (pkg.DIFFUSE, pkg.GLASS)
    assert abs(meshes[1].material[3] - 1.45) < 1e-6
    # the BVH is the BVH of the concatenated triangles
    whole = pkg.bvh_build(np.concatenate([a, b]))
    nodes = np.ctypeslib.as_array(C.cast(d.bvh_nodes, C.POINTER(C.c_float)), shape=(d.num_nodes, 16))
    ids = np.ctypeslib.as_array(C.cast(d.bvh_prim_ids, C.POINTER(C.c_uint32)), shape=(d.num_tri_float4,))
    assert np.array_equal(nodes.view(np.uint32), whole["nodes"].view(np.uint32))
    assert np.array_equal(ids, whole["prim_ids"])
    # a ray at the second mesh reports a face of the second mesh
    org = np.array([[1.9, 0.9, 6.0]], np.float32)
    dir_ = np.array([[0.013, 0.021, -1.0]], np.float32)
    dir_ /= np.linalg.norm(dir_)
    t, prim, _, _ = O.intersect(S.as_oracle_desc(scene), org, dir_)
    assert 316 <= prim[0] < 316 + 968
    one = pkg.HostScene(S.scene_file(pkg, "city"), 64, 48)
    assert one.desc.num_meshes == 0 and not one.desc.meshes


def test_obj_vertex_record_must_hold_three_coordinates(pkg, tmp_path):
    """A short `v x y` record is an error, not a silent shift of every later coordinate."""
    path = tmp_path / "short.obj"
    path.write_text("v 0 0 0\nv 1 0\nv 0 1 0\nv 0 0 1\nf 1 2 3\n")
    with pytest.raises(pkg.PtError, match="fewer than 3"):
        pkg.obj_load(str(path))
    path.write_text("v 0 0 0\nv 1 0 0   \nv 0 1 0\r\nf 1 2 3\n")
    assert pkg.obj_load(str(path)).shape == (1, 3, 3)
    # a face needs three corners, a coordinate has to be a finite number, an index has to name a vertex read so far
    for text, msg in [("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2\n", "fewer than 3 vertices"),
                      ("v 0 0 0\nv nan 0 0\nv 0 1 0\nf 1 2 3\n", "not a finite number"),
                      ("v 0 0 0\nv 1e999 0 0\nv 0 1 0\nf 1 2 3\n", "not a finite number"),
                      ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 9\n", "out of range"),
                      ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 -7\n", "out of range")]:
        path.write_text(text)
        with pytest.raises(pkg.PtError, match=msg):
            pkg.obj_load(str(path))
    path.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3")          # no newline at the end of the file
    assert pkg.obj_load(str(path)).shape == (1, 3, 3)


def ulp_error(got, want64):
    """Error of float32 results against float64 references, in units of the last place of the reference."""
    want32 = want64.astype(np.float32)
    spacing = np.spacing(np.maximum(np.abs(want32), np.float32(1e-30))).astype(np.float64)
    return np.abs(got.astype(np.float64) - want64) / spacing


def test_fixed_transcendental_kernels_are_accurate():
    """sin / cos / atan2 / acos / x^5 of the shading path are fixed polynomial kernels, stated once in the oracle and once
    in the device code, so that both take every random decision alike (the reference's CUDA math library, src/material.cuh:
    13-21,34-39 and src/environment_map.cuh:32-49, cannot be reproduced).  Here: their error against double precision over
    the ranges the path tracer uses."""
    rng = np.random.default_rng(5)
    x = np.concatenate([np.linspace(0, 2 * np.pi, 400001), rng.uniform(0, 2 * np.pi, 400000),
                        np.arange(0, 2049) * (2 * np.pi / 2048), np.arange(0, 1025) * (np.pi / 1024)]).astype(np.float32)
    s, c = O.math(0, x)
    x64 = x.astype(np.float64)
    assert np.abs(s - np.sin(x64)).max() < 1.5e-7 and np.abs(c - np.cos(x64)).max() < 1.5e-7
    assert np.abs(s * s + c * c - 1).max() < 4e-7
    y, xx = rng.normal(size=600000).astype(np.float32), rng.normal(size=600000).astype(np.float32)
    y[:8], xx[:8] = [0, 0, 1, -1, 0, 0, 1e-30, -1e-30], [1, -1, 0, 0, 0, -0.0, 1e30, -1e30]
    a = O.math(1, y, xx)
    want = np.arctan2(y.astype(np.float64), xx.astype(np.float64))
    assert np.abs(a[8:] - want[8:]).max() < 4e-7               # |atan2| <= pi: 1.7 ulp at pi
    assert ulp_error(a[8:], want[8:]).max() <= 3.0
    # the axes and the origin, with the zeros' signs treated as atan2f treats them: atan2(0, 0) = 0, atan2(0, -0) = pi
    assert a[0] == 0 and abs(a[1] - np.pi) < 3e-7 and abs(a[2] - np.pi / 2) < 2e-7 and abs(a[3] + np.pi / 2) < 2e-7
    assert a[4] == 0 and abs(a[5] - np.pi) < 3e-7 and abs(a[6]) < 1e-37 and abs(a[7] + np.pi) < 3e-7
    # signed zeros: the sign of y decides the sign of the result, -0 included (atan2f(-0, x < 0) = -pi)
    z = O.math(1, np.array([-0.0, -0.0, 0.0, -0.0], np.float32), np.array([-1.0, 1.0, -0.0, -0.0], np.float32))
    assert abs(z[0] + np.pi) < 3e-7 and z[1] == 0 and np.signbit(z[1]) and abs(z[2] - np.pi) < 3e-7 and abs(z[3] + np.pi) < 3e-7
    v = np.concatenate([np.linspace(-1, 1, 400001), rng.uniform(-1, 1, 200000), 1 - np.geomspace(1e-7, 1, 2000)]).astype(np.float32)
    ac = O.math(2, v)
    want = np.arccos(v.astype(np.float64))
    assert np.abs(ac - want).max() < 4e-7
    big = want > 1e-3
    assert ulp_error(ac[big], want[big]).max() <= 3.0
    assert ac[v == 1].max() == 0 and abs(ac[v == -1][0] - np.pi) < 3e-7
    p = O.math(3, v)
    assert ulp_error(p, v.astype(np.float64) ** 5).max() <= 3.0          # three roundings


def test_device_and_oracle_state_the_same_transcendental_kernels():
    """csrc/pt_math.h (what the kernels evaluate), compiled for the host, returns the oracle's bits on every input tried:
    the two statements of the polynomial kernels are the same function.  (That the device evaluates it to the same bits
    as the host is what the bit-equal films of the GPU tests show.)"""
    lib = C.CDLL(os.path.join(ROOT, "cuda-path-tracer_amd", "libpt_math_host.so"))
    lib.ptm_eval.argtypes = [C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ptm_eval.restype = None
    rng = np.random.default_rng(11)

    def device_header(which, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(a if b is None else b, dtype=np.float32)
        o0, o1 = np.zeros_like(a), np.zeros_like(a)
        lib.ptm_eval(which, a.size, O.ptr(a), O.ptr(b), O.ptr(o0), O.ptr(o1))
        return (o0, o1) if which == 0 else o0

    def same(x, y):
        return np.array_equal(x.view(np.uint32), y.view(np.uint32))

    turn = np.concatenate([rng.uniform(0, 2 * np.pi, 1000000), np.linspace(0, 6.2831855, 200001)]).astype(np.float32)
    (s0, c0), (s1, c1) = device_header(0, turn), O.math(0, turn)
    assert same(s0, s1) and same(c0, c1)
    y, x = rng.normal(size=1000000).astype(np.float32), rng.normal(size=1000000).astype(np.float32)
    y[:6], x[:6] = [0, 0, 1, -1, 0, -0.0], [1, -1, 0, 0, 0, -0.0]
    assert same(device_header(1, y, x), O.math(1, y, x))
    unit = np.concatenate([rng.uniform(-1, 1, 1000000), [-1, 1, 0, 0.5, -0.5]]).astype(np.float32)
    assert same(device_header(2, unit), O.math(2, unit))
    assert same(device_header(3, unit), O.math(3, unit))
