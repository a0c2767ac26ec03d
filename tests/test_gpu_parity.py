"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle and the frozen fixtures.

Bars: primitive ids bit-exact; hit distance t within 1e-5 relative (the tests in fact demand bit equality, which the
fixed fma sequence delivers, of the hit normals too); camera rays bit-exact; films bit-exact (same random numbers, same operation sequence, see assert_same_film).
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
import scenes as S

pytestmark = pytest.mark.gpu

GOLDEN = S.GOLDEN
T_REL_TOL = 1e-5          # the stated bar for hit distances


def check_hits(t, prim, ot, oprim, what):
    assert np.array_equal(prim, oprim), "%s: %d primitive ids differ" % (what, int((prim != oprim).sum()))
    hit = oprim != O.PRIM_MISS
    rel = np.abs(t[hit] - ot[hit]) / np.maximum(np.abs(ot[hit]), 1e-30)
    assert rel.size == 0 or rel.max() <= T_REL_TOL, "%s: t differs by %g relative" % (what, rel.max())
    assert np.array_equal(t[~hit], ot[~hit]), "%s: miss distance is not FLT_MAX" % what
    return bool(np.array_equal(t, ot))


@pytest.mark.parametrize("name", ["knot", "balls", "bunny_lowpoly", "suzanne"])
def test_intersect_matches_golden_fixture(pkg, ctx, name):
    scene = S.parity_cases(pkg)[name]
    g = np.load(os.path.join(GOLDEN, "rays_%s.npz" % name))
    ctx.upload(scene)
    t, prim, _ = ctx.intersect(g["org"], g["dir"])
    bit_equal = check_hits(t, prim, g["t"], g["prim"], name)
    assert bit_equal, "%s: t is within tolerance but not bit-equal to the fixture" % name


@pytest.mark.parametrize("mesh,cam", [("bunny", ((1, 1, 4), (-1, 0.5, 0))), ("city", ((-10, 12, 10), (0, 2, 0)))])
def test_intersect_matches_oracle_on_shipped_meshes(pkg, ctx, mesh, cam):
    """Primary + one-bounce + axis-parallel rays on the reference's larger meshes (28,576 and 14,350 triangles)."""
    W, H = 160, 96
    scene = S.mesh_scene(pkg, S.golden_mesh(mesh), W, H, cam[0], cam[1], fov=38.0,
                         planes=[pkg.make_plane((0, 0, 0), (0, 1, 0), (0.3, 0.3, 0.3, 0), pkg.DIFFUSE)],
                         spheres=[pkg.make_sphere((-3, 1.5, -0.5), 1.5, (0.5, 0.5, 0.5, 0), pkg.MIRROR)])
    org, dir_ = S.ray_batch(scene, W, H, seed=3)
    ctx.upload(scene)
    t, prim, nrm = ctx.intersect(org, dir_, want_normals=True)
    ot, oprim, onrm, _ = O.intersect(S.as_oracle_desc(scene), org, dir_, want_normals=True)
    check_hits(t, prim, ot, oprim, mesh)
    assert np.array_equal(t, ot)
    assert np.array_equal(nrm.view(np.uint32), onrm.view(np.uint32)), "hit normals differ from the oracle's in some bit"
    assert (prim != O.PRIM_MISS).mean() > 0.3


def test_instanced_stress_scene_matches_oracle(pkg, ctx):
    """The 8-GPU stress configuration in small: 9 copies of the 9,216-triangle knot flattened into one glossy mesh with
    glass spheres between them.  Hits and film bit-equal to the oracle."""
    W, H, spp = 128, 72, 4
    scene, ntris = S.instanced_scene(pkg, W, H, 9, base="knot_small.obj")
    assert ntris == 9 * 9216 and scene.desc.num_spheres == 4
    org, dir_ = S.ray_batch(scene, W, H, seed=8)
    ctx.upload(scene)
    t, prim, _ = ctx.intersect(org, dir_)
    ot, oprim, _, _ = O.intersect(S.as_oracle_desc(scene), org, dir_)
    check_hits(t, prim, ot, oprim, "instanced")
    assert np.array_equal(t, ot)
    assert ctx.env_is_packed                      # the forest map comes from an .hdr file
    ctx.film_bind(W, H)
    ctx.render(0, spp, spp, seed=4)
    film = ctx.film_download()
    ofilm, _ = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=4)
    assert_same_film(film, ofilm, "instanced")


def test_device_reproduces_handworked_hits(pkg, ctx):
    """Closest hits that follow by hand (tests/scenes.py::handworked_hits), not from the oracle: the device returns exactly
    those distances, ids and normals, with both traversal kernels."""
    scene, org, dir_, t, prim, nrm = S.handworked_hits(pkg)
    ctx.upload(scene)
    try:
        for mode in (2, 1):
            ctx.set_option(pkg.OPT_EXTEND_MODE, mode)
            gt, gprim, gnrm = ctx.intersect(org, dir_, want_normals=True)
            assert np.array_equal(gprim, prim), mode
            assert np.array_equal(gt, t), mode
            hit = prim != O.PRIM_MISS
            assert np.array_equal(gnrm[hit], nrm[hit]), mode
    finally:
        ctx.set_option(pkg.OPT_EXTEND_MODE, 2)


def test_device_reproduces_handworked_film(pkg, ctx):
    """A film that follows by hand (tests/scenes.py::handworked_film), not from the oracle: the device's film holds exactly
    the environment's radiance in the sky pixels and albedo times it in the mirror floor's, in both render modes."""
    scene, want = S.handworked_film(pkg)
    known = ~np.isnan(want)
    H, W = want.shape[:2]
    ctx.upload(scene)
    ctx.film_bind(W, H)
    try:
        for mode in (1, 2, 0):
            ctx.set_option(pkg.OPT_RENDER_MODE, mode)
            for spp in (1, 4, 64):
                ctx.film_clear()
                st = ctx.render(0, spp, spp, seed=spp)
                film = ctx.film_download()
                assert np.array_equal(film[known], want[known]), (mode, spp)
                assert st.paths == W * H * spp
    finally:
        ctx.set_option(pkg.OPT_RENDER_MODE, 0)


def test_device_converges_to_radiometry(pkg, ctx):
    """Films known from radiometry alone (tests/scenes.py::furnace_cases): the device's mean over the film, 8.4 M paths per
    case (statistical error about 0.02 %), is within 1 % of the float64 integral resp. the closed form, in both render modes;
    what remains is the bias of the reference's table-driven light sampler."""
    try:
        for name, (scene, W, H, want) in S.furnace_cases(pkg).items():
            ctx.upload(scene)
            ctx.film_bind(W, H)
            for mode in (1, 2):
                ctx.set_option(pkg.OPT_RENDER_MODE, mode)
                ctx.film_clear()
                st = ctx.render(0, 16384, 16384, seed=5)
                got = ctx.film_download().reshape(-1, 3).mean(axis=0, dtype=np.float64)
                # camera -> surface -> sky; on the mesh a rare grazing ray next to an edge meets the neighbouring facet too
                assert 0 <= st.rays - 2 * st.paths <= 1e-5 * st.paths or name in ("clear glass sphere", "hall of mirrors"), \
                    (name, mode)
                assert np.abs(got / want - 1).max() < 0.01, (name, mode, got, want)
    finally:
        ctx.set_option(pkg.OPT_RENDER_MODE, 0)


def test_device_follows_the_fresnel_curve(pkg, ctx):
    """tests/scenes.py::fresnel_case: every pixel of the device's film of a glass plane between two sky colours, 16,384
    samples each, is within 1.5 % of the colour difference of R A + (1 - R) B with Schlick's R at that pixel's elevation
    (statistical error 0.4 %), and the film mean within 0.1 %, in both render modes."""
    scene, want, span = S.fresnel_case(pkg)
    H, W = want.shape[:2]
    ctx.upload(scene)
    ctx.film_bind(W, H)
    try:
        for mode in (1, 2):
            ctx.set_option(pkg.OPT_RENDER_MODE, mode)
            ctx.film_clear()
            st = ctx.render(0, 16384, 16384, seed=9)
            film = ctx.film_download().astype(np.float64)
            assert st.rays == 2 * st.paths, mode
            assert np.abs(film - want).max() < 0.015 * span, (mode, np.abs(film - want).max() / span)
            assert np.abs(film.mean(axis=(0, 1)) - want.mean(axis=(0, 1))).max() < 0.001 * span, mode
    finally:
        ctx.set_option(pkg.OPT_RENDER_MODE, 0)


def test_device_shades_a_sphere_as_radiometry_says(pkg, ctx):
    """tests/scenes.py::shaded_sphere_case: every pixel of the device's film (16,384 samples) is within 3 % and every 4 x 4
    block within 1 % of rho / pi x the cosine-weighted integral of the sky around that pixel's normal, in both render
    modes."""
    scene, want = S.shaded_sphere_case(pkg)
    H, W = want.shape[:2]
    blocks = lambda a: a.reshape(H // 4, 4, W // 4, 4, 3).mean(axis=(1, 3), dtype=np.float64)
    ctx.upload(scene)
    ctx.film_bind(W, H)
    try:
        for mode in (1, 2):
            ctx.set_option(pkg.OPT_RENDER_MODE, mode)
            ctx.film_clear()
            st = ctx.render(0, 16384, 16384, seed=11)
            film = ctx.film_download().astype(np.float64)
            assert st.rays == 2 * st.paths, mode
            assert np.abs(film / want - 1).max() < 0.03, (mode, np.abs(film / want - 1).max())
            assert np.abs(blocks(film) / blocks(want) - 1).max() < 0.01, (mode, np.abs(blocks(film) / blocks(want) - 1).max())
    finally:
        ctx.set_option(pkg.OPT_RENDER_MODE, 0)


def test_both_extend_kernels_give_identical_hits(pkg, ctx):
    """The loop-per-phase batch kernel and the persistent single-step kernel must agree bit for bit with the fixture."""
    scene = S.parity_cases(pkg)["knot"]
    g = np.load(os.path.join(GOLDEN, "rays_knot.npz"))
    ctx.upload(scene)
    reps = 40                                   # enough rays that persistent waves refill many times
    org = np.tile(g["org"], (reps, 1))
    dir_ = np.tile(g["dir"], (reps, 1))
    try:
        for mode in (1, 2):
            ctx.set_option(pkg.OPT_EXTEND_MODE, mode)
            t, prim, _ = ctx.intersect(org, dir_)
            assert np.array_equal(prim, np.tile(g["prim"], reps)), "mode %d" % mode
            assert np.array_equal(t, np.tile(g["t"], reps)), "mode %d" % mode
    finally:
        ctx.set_option(pkg.OPT_EXTEND_MODE, 2)


def test_traversal_tuning_changes_no_answer(pkg, ctx):
    """The traversal stack's split (LDS column + overflow in global memory for what a deep tree needs beyond it), the
    refill threshold, the blocks per CU, the chunk order and the node order at upload are tuning: with any of them the hits
    must equal the fixture bit for bit and the film must not change by a bit.  4 LDS slots on a tree 16 deep sends most of
    every ray's stack through the overflow."""
    W, H, spp = 96, 64, 4
    scene = S.parity_cases(pkg)["knot"]
    g = np.load(os.path.join(GOLDEN, "rays_knot.npz"))
    reps = 20
    org, dir_ = np.tile(g["org"], (reps, 1)), np.tile(g["dir"], (reps, 1))
    want_prim, want_t = np.tile(g["prim"], reps), np.tile(g["t"], reps)

    def answers():
        ctx.upload(scene)                       # the node order is applied at upload
        t, prim, _ = ctx.intersect(org, dir_)
        ctx.film_bind(W, H)
        ctx.render(0, spp, spp, seed=21)
        return t, prim, ctx.film_download()

    ctx.set_option(pkg.OPT_RENDER_MODE, 1)            # the options below tune the wavefront loop's traversal kernel
    _, _, film0 = answers()
    defaults = {pkg.OPT_TUNE_LDS_SLOTS: 26, pkg.OPT_TUNE_REFILL_IDLE: 44, pkg.OPT_TUNE_EXTEND_BLOCKS: 8,
                pkg.OPT_TUNE_XCD_GROUP: 0, pkg.OPT_NODE_LAYOUT: 1}
    cases = [(pkg.OPT_TUNE_LDS_SLOTS, 4), (pkg.OPT_TUNE_LDS_SLOTS, 2), (pkg.OPT_TUNE_REFILL_IDLE, 1),
             (pkg.OPT_TUNE_REFILL_IDLE, 64), (pkg.OPT_TUNE_EXTEND_BLOCKS, 1), (pkg.OPT_TUNE_XCD_GROUP, 1),
             (pkg.OPT_NODE_LAYOUT, 0)]
    try:
        for opt, value in cases:
            ctx.set_option(opt, value)
            t, prim, film = answers()
            ctx.set_option(opt, defaults[opt])
            assert np.array_equal(prim, want_prim), "option %d = %d: primitive ids" % (opt, value)
            assert np.array_equal(t, want_t), "option %d = %d: hit distances" % (opt, value)
            assert np.array_equal(film, film0), "option %d = %d: film" % (opt, value)
    finally:
        for opt, value in defaults.items():
            ctx.set_option(opt, value)
        ctx.set_option(pkg.OPT_RENDER_MODE, 0)


def test_primary_rays_bit_exact(pkg, ctx):
    """generate stage: same counter-based random numbers, same arithmetic -> identical camera rays (thin lens on)."""
    for W, H, pinhole in [(48, 32, False), (50, 30, False), (48, 32, True)]:      # tiled and untiled pixel enumeration
        # reference scene/suzanne.scene has aperture 0: the pinhole branch of src/camera.cuh:41-44
        scene = pkg.HostScene(S.scene_file(pkg, "suzanne"), W, H) if pinhole else S.spheres_scene(pkg, W, H)
        ctx.upload(scene)
        for sample in (0, 5):
            org, dir_ = ctx.primary_rays(W, H, sample=sample, seed=99)
            oorg, odir = O.primary_rays(S.as_oracle_desc(scene), W, H, sample=sample, seed=99)
            assert np.array_equal(org, oorg)
            assert np.array_equal(dir_, odir)


def test_device_evaluates_the_transcendental_kernels_to_the_oracles_bits(pkg, ctx):
    """sin / cos / atan2 / acos / x^5 as the device evaluates them against the oracle's statement of the same kernels, on
    millions of inputs over and beyond the ranges the path tracer feeds them (signed zeros, the axes, the ends of the
    intervals included): every result the same bits."""
    rng = np.random.default_rng(19)

    def same(x, y):
        return np.array_equal(x.view(np.uint32), y.view(np.uint32))

    turn = np.concatenate([rng.uniform(0, 2 * np.pi, 2000000), np.linspace(0, 6.2831855, 500001),
                           np.arange(0, 2049) * (2 * np.pi / 2048), np.arange(0, 1025) * (np.pi / 1024)]).astype(np.float32)
    (s0, c0), (s1, c1) = ctx.math_eval(0, turn), O.math(0, turn)
    assert same(s0, s1) and same(c0, c1)
    y, x = rng.normal(size=2000000).astype(np.float32), rng.normal(size=2000000).astype(np.float32)
    y[:10], x[:10] = [0, 0, 1, -1, 0, -0.0, 1e-30, -1e-30, 1e38, 3], [1, -1, 0, 0, 0, -0.0, 1e30, -1e30, 1e38, 3]
    assert same(ctx.math_eval(1, y, x), O.math(1, y, x))
    unit = np.concatenate([rng.uniform(-1, 1, 2000000), np.linspace(-1, 1, 500001), 1 - np.geomspace(1e-7, 1, 5000),
                           [-1, 1, 0, -0.0, 0.5, -0.5]]).astype(np.float32)
    assert same(ctx.math_eval(2, unit), O.math(2, unit))
    assert same(ctx.math_eval(3, unit), O.math(3, unit))


def test_empty_and_single_ray(pkg, ctx):
    scene = S.spheres_scene(pkg, 48, 32)
    ctx.upload(scene)
    t, prim, _ = ctx.intersect(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
    assert t.shape == (0,) and prim.shape == (0,)
    t, prim, _ = ctx.intersect(np.array([[0, 1, 5]], np.float32), np.array([[0, -0.2, -1]], np.float32) / np.float32(1.0198039))
    ot, oprim, _, _ = O.intersect(S.as_oracle_desc(scene), np.array([[0, 1, 5]], np.float32),
                                  np.array([[0, -0.2, -1]], np.float32) / np.float32(1.0198039))
    assert np.array_equal(prim, oprim) and np.array_equal(t, ot)


def film_metrics(a, b):
    """relMSE = mean((a-b)^2 / (b^2 + 1e-2)) over channels, and the share of pixels whose largest channel difference
    exceeds 1e-3 of the pixel's magnitude."""
    d = a.astype(np.float64) - b.astype(np.float64)
    rel_mse = float(np.mean(d * d / (b.astype(np.float64) ** 2 + 1e-2)))
    differing = float((np.abs(d).max(axis=2) > 1e-3 * (1e-3 + np.abs(b).max(axis=2))).mean())
    return rel_mse, differing


# Film bar against the oracle at equal spp and equal seed: BIT EQUALITY, and equal ray counts.  Both sides draw the same
# counter-based random numbers, execute the same fma sequence and evaluate sin / cos / atan2 / acos / x^5 with the same
# fixed polynomial kernels (csrc/pt_math.h; the oracle states them separately), and the slab is summed in sample order
# as the oracle accumulates: every path takes the same decisions and contributes the same floats.  (With a math
# library's transcendentals, whose last place differs between device and host, about 0.07 % of the paths used to flip a
# discrete decision.)  Against the reference's own renders the bar is statistical, see test_converged_render_matches_oracle.
def assert_same_film(film, ofilm, what=""):
    # a NaN (the 0/0 of reference src/render.cu:145-151 when both pdfs vanish) must be a NaN on both sides
    differing = ((film.view(np.uint32) != ofilm.view(np.uint32)) & ~(np.isnan(film) & np.isnan(ofilm))).any(axis=2)
    assert not differing.any(), "%s: %d of %d pixels differ from the oracle in some bit (max abs %g)" % (
        what, int(differing.sum()), differing.size, float(np.abs(film - ofilm).max()))


@pytest.mark.parametrize("name", ["knot", "balls", "city", "suzanne"])
def test_render_matches_oracle(pkg, ctx, name):
    W, H, spp = 96, 64, 8
    if name in ("city", "suzanne"):                  # reference scene/city.scene, scene/suzanne.scene with the shipped meshes
        scene = pkg.HostScene(S.scene_file(pkg, name), W, H)
    else:
        scene = S.knot_scene(pkg, W, H) if name == "knot" else S.spheres_scene(pkg, W, H)
    ctx.upload(scene)
    ctx.film_bind(W, H)
    st = ctx.render(0, spp, spp, seed=5)
    film = ctx.film_download()
    ofilm, ost = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=5)
    assert st.paths == W * H * spp == ost.paths
    assert st.rays == ost.rays
    assert_same_film(film, ofilm, name)
    # and the 16-bit image a user sees: identical through the host tone map
    assert np.array_equal(pkg.tonemap16(film), pkg.tonemap16(ofilm))


# The two benchmark configurations with a mesh and an .hdr map, loaded from their scene files exactly as the benchmark
# and the command line load them: reference scene/dragon.scene (868,352-triangle stand-in for the unshipped dragon.obj,
# forest.hdr stand-in) and scene/bunny.scene (the shipped bunny.obj, sunset.hdr stand-in with a small sun disc of radiance
# 5e3, the case where the light-sampling pdf of reference src/render.cu:145-151 is most extreme).
# dragon_dense: the same scene file with the frame-filling variant of the stand-in mesh (bench.py --scene dragon_dense).
@pytest.mark.parametrize("name", ["dragon", "bunny", "dragon_dense"])
def test_benchmark_scene_files_match_oracle(pkg, ctx, name):
    W, H, spp = 96, 54, 8
    scene = pkg.HostScene(S.scene_file(pkg, name), W, H)
    assert scene.desc.num_tris == {"dragon": 868352, "bunny": 28576, "dragon_dense": 868352}[name]
    ctx.upload(scene)
    assert ctx.env_is_packed
    # (a) closest hits on primary + bounced + axis-parallel rays: bit-equal
    org, dir_ = S.ray_batch(scene, W, H, seed=12)
    t, prim, nrm = ctx.intersect(org, dir_, want_normals=True)
    ot, oprim, onrm, _ = O.intersect(S.as_oracle_desc(scene), org, dir_, want_normals=True)
    check_hits(t, prim, ot, oprim, name)
    assert np.array_equal(t, ot)
    assert np.array_equal(nrm.view(np.uint32), onrm.view(np.uint32)), "hit normals differ from the oracle's in some bit"
    assert (prim != O.PRIM_MISS).mean() > 0.3
    # (b) a small render against the oracle: bit-equal
    ctx.film_bind(W, H)
    st = ctx.render(0, spp, spp, seed=5)
    film = ctx.film_download()
    ofilm, ost = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=5)
    assert st.paths == W * H * spp == ost.paths
    assert st.rays == ost.rays
    assert np.isfinite(ofilm).all() and np.isfinite(film).all() and (film >= 0).all()
    assert_same_film(film, ofilm, name)
    if name == "dragon_dense":
        # what makes it the traversal-bound workload: the mesh fills the frame and a ray visits many nodes
        po, pd = O.primary_rays(S.as_oracle_desc(scene), W, H, sample=0, seed=1)
        _, pprim, _, _ = O.intersect(S.as_oracle_desc(scene), po, pd)
        assert (pprim < O.PRIM_SPHERE).mean() >= 0.8
        assert ost.nodes / ost.rays >= 40


def test_readme_configuration_matches_oracle(pkg, ctx):
    """The configuration the reference's README times (README.md:76): scene/balls.scene as written (one plane, four spheres,
    all four materials, thin lens, the 2048x1024 studio map stand-in) at 960x540 with 32 samples per pixel, loaded from the
    scene file as the command line loads it.  Film and ray count bit-equal to the oracle's."""
    W, H, spp = 960, 540, 32
    scene = pkg.HostScene(S.scene_file(pkg, "balls"), W, H)
    assert scene.desc.num_planes == 1 and scene.desc.num_spheres == 4 and scene.desc.num_nodes == 0
    assert (scene.desc.env_width, scene.desc.env_height) == (2048, 1024)
    ctx.upload(scene)
    ctx.film_bind(W, H)
    st = ctx.render(0, spp, spp, seed=0)
    film = ctx.film_download()
    ofilm, ost = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=0, nthreads=16)
    assert st.paths == W * H * spp == ost.paths == 16588800
    assert st.rays == ost.rays
    assert np.isfinite(film).all()
    assert_same_film(film, ofilm, "balls.scene 960x540x32")


def test_4k_frame_in_three_slabs(pkg, ctx):
    """A 3840x2160 frame of the dragon scene (the resolution of the largest benchmark configuration) with a sample slab that
    holds two of its six samples per pixel: the call runs three slabs, starts every path once, traces the same rays and
    gives the one-slab film up to the order in which a pixel's samples are added."""
    W, H, spp = 3840, 2160, 6
    scene = pkg.HostScene(S.scene_file(pkg, "dragon"), W, H)
    ctx.upload(scene)
    ctx.film_bind(W, H)
    try:
        one = ctx.render(0, spp, spp, seed=2)
        film_one = ctx.film_download()
        ctx.film_clear()
        ctx.set_option(pkg.OPT_SLAB_BYTES, 2 * W * H * 16)
        three = ctx.render(0, spp, spp, seed=2)
        film_three = ctx.film_download()
    finally:
        ctx.set_option(pkg.OPT_SLAB_BYTES, 32 << 30)
    assert one.paths == three.paths == W * H * spp
    assert one.rays == three.rays
    assert np.isfinite(film_three).all() and (film_three >= 0).all()
    assert np.allclose(film_three, film_one, rtol=1e-5, atol=1e-6)
    print("4K x %d spp: one slab %.1f ms, three slabs %.1f ms" % (spp, one.ms_total, three.ms_total))


@pytest.mark.parametrize("name", ["dragon", "bunny", "dragon_dense", "city"])
def test_benchmark_scene_files_at_full_size(pkg, ctx, name):
    """(c) 1920x1080 at 2 spp, the benchmark's frame size: every path started once, film and ray count bit-equal to the
    oracle's, the film finite and non-negative (a 0/0 from reference src/render.cu:145-151 would show here), the samples split
    over two calls equal to one call up to float addition order, and the ray count the sum over both halves."""
    W, H, spp = 1920, 1080, 2
    scene = pkg.HostScene(S.scene_file(pkg, name), W, H)
    ctx.upload(scene)
    ctx.film_bind(W, H)
    st = ctx.render(0, spp, spp, seed=1)
    full = ctx.film_download()
    ofilm, ost = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=1, nthreads=16)
    assert st.rays == ost.rays
    assert_same_film(full, ofilm, "%s 1920x1080" % name)
    assert st.paths == W * H * spp
    assert np.isfinite(full).all() and (full >= 0).all()
    assert full.mean() > 1e-3
    acc = np.zeros_like(full)
    rays = 0
    for r in range(2):
        b, c = pkg.spp_partition(spp, 2, r)
        ctx.film_clear()
        s2 = ctx.render(b, c, spp, seed=1)
        rays += s2.rays
        acc += ctx.film_download()
    assert rays == st.rays
    assert np.allclose(acc, full, rtol=1e-5, atol=1e-6)


def test_many_and_coincident_primitives(pkg, ctx):
    """Planes and spheres are tested in the order given with a shrinking t_max and a strict comparison, so of two
    coincident primitives the earlier one is hit (reference src/render.cu:25-33).  40 spheres, 3 planes, two coincident
    pairs with different materials, a camera inside a glass sphere, over a mesh: primitive ids, t and film bit-equal to
    the oracle."""
    W, H, spp = 96, 60, 4
    rng = np.random.default_rng(41)
    planes = [pkg.make_plane((0, 0, 0), (0, 1, 0), (0.4, 0.4, 0.4, 0), pkg.DIFFUSE),
              pkg.make_plane((0, 0, 0), (0, 1, 0), (0.9, 0.9, 0.9, 0), pkg.MIRROR),        # coincident with the first: never hit
              pkg.make_plane((0, 0, -6), (0, 0.2, 1), (0.5, 0.5, 0.6, 0.05), pkg.GLOSSY)]
    spheres = [pkg.make_sphere((0, 1.2, 5), 0.8, (1, 1, 1, 1.5), pkg.GLASS)]               # the camera sits inside this one
    types = [pkg.DIFFUSE, pkg.MIRROR, pkg.GLOSSY, pkg.GLASS]
    for k in range(38):
        c = (float(rng.uniform(-4, 4)), float(rng.uniform(0.2, 2.5)), float(rng.uniform(-4, 2)))
        spheres.append(pkg.make_sphere(c, float(rng.uniform(0.15, 0.5)), (0.8, 0.6, 0.5, 1.4 if k % 4 == 3 else 0.05), types[k % 4]))
    spheres.append(pkg.make_sphere((1.5, 0.6, 1.0), 0.6, (0.2, 0.8, 0.2, 0), pkg.DIFFUSE))
    spheres.append(pkg.make_sphere((1.5, 0.6, 1.0), 0.6, (1, 1, 1, 0), pkg.MIRROR))        # coincident with the one before
    bvh = pkg.bvh_build(S.knot_mesh(pkg))
    cam = pkg.camera_make((0, 1.2, 5), (0, 0.8, 0), 45.0, W / H, 0.08, 5.0)
    tex, marg, cond = S.small_env(pkg)
    scene = pkg.ManualScene(cam, tex, marg, cond, planes=planes, spheres=spheres, bvh=bvh,
                            mesh_material=(0.6, 0.3, 0.1, 0.0), mesh_material_type=pkg.DIFFUSE)
    org, dir_ = S.ray_batch(scene, W, H, seed=31)
    ctx.upload(scene)
    t, prim, nrm = ctx.intersect(org, dir_, want_normals=True)
    ot, oprim, onrm, _ = O.intersect(S.as_oracle_desc(scene), org, dir_, want_normals=True)
    check_hits(t, prim, ot, oprim, "many primitives")
    assert np.array_equal(t, ot) and np.array_equal(nrm.view(np.uint32), onrm.view(np.uint32))
    hit_planes = set(int(p & 0x3FFFFFFF) for p in prim[(prim & 0x80000000) != 0] if p != O.PRIM_MISS)
    hit_spheres = set(int(p & 0x3FFFFFFF) for p in prim[((prim & 0x40000000) != 0) & ((prim & 0x80000000) == 0)])
    assert 0 in hit_planes and 1 not in hit_planes
    assert 39 in hit_spheres and 40 not in hit_spheres and len(hit_spheres) > 20
    ctx.film_bind(W, H)
    st = ctx.render(0, spp, spp, seed=31)
    ofilm, ost = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=31)
    assert st.rays == ost.rays
    assert_same_film(ctx.film_download(), ofilm, "many primitives")


def test_depth_limit_in_a_hall_of_mirrors(pkg, ctx):
    """Two facing mirror planes of albedo 1 and a glass sphere between them: Russian roulette never ends a path (its
    survival probability is the throughput's largest channel, reference src/render.cu:78-86), so almost every path runs
    into the depth cut of src/render.cu:62 (two increments per bounce: 8 bounces) and contributes through it.  Film and
    ray count bit-equal to the oracle."""
    W, H, spp = 80, 48, 6
    cam = pkg.camera_make((0, 1, 4), (0, 1, 0), 50.0, W / H, 0.05, 4.0)
    tex, marg, cond = S.small_env(pkg)
    planes = [pkg.make_plane((0, 0, 0), (0, 1, 0), (1, 1, 1, 0), pkg.MIRROR),
              pkg.make_plane((0, 2, 0), (0, -1, 0), (1, 1, 1, 0), pkg.MIRROR)]
    spheres = [pkg.make_sphere((0, 1, 0), 0.6, (1, 1, 1, 1.5), pkg.GLASS)]
    scene = pkg.ManualScene(cam, tex, marg, cond, planes=planes, spheres=spheres)
    ctx.upload(scene)
    ctx.film_bind(W, H)
    st = ctx.render(0, spp, spp, seed=17)
    film = ctx.film_download()
    ofilm, ost = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=17)
    assert st.paths == W * H * spp and st.rays == ost.rays
    assert st.rays > 7 * st.paths                      # nine rays for a path that reaches the cut
    assert_same_film(film, ofilm, "hall of mirrors")
    assert film.mean() > 0.05                          # the cut paths do contribute


def test_scene_with_two_meshes_matches_oracle(pkg, ctx):
    """Two OBJ blocks with different materials (Lambertian bunny, glass suzanne) in one BVH: hits bit-equal to the oracle,
    the film bit-equal too, and the per-mesh material visibly in effect (the same triangles with one material
    for all give another film)."""
    W, H, spp = 96, 64, 8
    scene = pkg.HostScene(S.scene_file(pkg, "two_meshes"), W, H)
    assert scene.desc.num_meshes == 2
    ctx.upload(scene)
    org, dir_ = S.ray_batch(scene, W, H, seed=15)
    t, prim, nrm = ctx.intersect(org, dir_, want_normals=True)
    ot, oprim, onrm, _ = O.intersect(S.as_oracle_desc(scene), org, dir_, want_normals=True)
    check_hits(t, prim, ot, oprim, "two meshes")
    assert np.array_equal(t, ot)
    assert np.array_equal(nrm.view(np.uint32), onrm.view(np.uint32))
    assert ((prim < 316)).any() and ((prim >= 316) & (prim < 316 + 968)).any()
    ctx.film_bind(W, H)
    st = ctx.render(0, spp, spp, seed=5)
    film = ctx.film_download()
    ofilm, ost = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=5)
    assert st.rays == ost.rays
    assert_same_film(film, ofilm, "two meshes")
    # the same buffers as ONE mesh (the first material for every triangle): a different image
    one = pkg.SceneDesc.from_buffer_copy(scene.desc)
    one.meshes, one.num_meshes = None, 0
    ctx.upload(one)
    ctx.film_bind(W, H)
    ctx.render(0, spp, spp, seed=5)
    assert film_metrics(ctx.film_download(), film)[1] > 0.02
    # a mesh list that does not start at face 0, or names an unknown material, is refused
    bad = (pkg.Mesh * 2)(pkg.Mesh((C.c_float * 4)(1, 1, 1, 0), pkg.DIFFUSE, 5), pkg.Mesh((C.c_float * 4)(1, 1, 1, 0), pkg.GLASS, 316))
    d2 = pkg.SceneDesc.from_buffer_copy(scene.desc)
    d2.meshes, d2.num_meshes = C.addressof(bad), 2
    with pytest.raises(pkg.PtError, match="first_face"):
        ctx.upload(d2)
    bad[0].first_face, bad[1].material_type = 0, 3
    with pytest.raises(pkg.PtError, match="invalid material"):
        ctx.upload(d2)


def test_render_matches_frozen_film(pkg, ctx):
    """Bit equality with the oracle film frozen in tests/golden/film_knot.npz (no oracle run involved)."""
    g = np.load(os.path.join(GOLDEN, "film_knot.npz"))
    W, H, spp, seed = int(g["width"]), int(g["height"]), int(g["spp"]), int(g["seed"])
    scene = S.knot_scene(pkg, W, H)
    ctx.upload(scene)
    ctx.film_bind(W, H)
    st = ctx.render(0, spp, spp, seed=seed)
    film = ctx.film_download()
    assert st.paths == int(g["paths"]) and st.rays == int(g["rays"])
    assert_same_film(film, g["film"], "frozen film")


def test_converged_render_matches_oracle(pkg, ctx):
    """A small frame taken close to convergence (1024 spp).  Against the oracle with the same seed: bit-equal.  Against
    the oracle with ANOTHER seed -- the situation of a comparison with the reference's own renders, whose random stream
    (cuRAND) is not ours -- the stated image tolerance: relMSE of the linear film and RMSE of the tone-mapped image at most
    1.5 times what two oracle renders with different seeds differ by, and the tone-mapped RMSE below 2 % of full scale."""
    W, H, spp = 64, 48, 1024
    scene = S.knot_scene(pkg, W, H)
    ctx.upload(scene)
    ctx.film_bind(W, H)
    ctx.render(0, spp, spp, seed=2)
    film = ctx.film_download()
    ofilm, _ = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=2)
    assert_same_film(film, ofilm, "1024 spp")
    other, _ = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=3)
    third, _ = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=4)
    tm = lambda f: pkg.tonemap16(f).astype(np.float64) / 65535
    rmse = np.sqrt(np.mean((tm(film) - tm(other)) ** 2))
    rmse_floor = np.sqrt(np.mean((tm(third) - tm(other)) ** 2))
    rel_mse, floor = film_metrics(film, other)[0], film_metrics(third, other)[0]
    assert rmse <= 1.5 * rmse_floor and rmse <= 0.02, "tone-mapped RMSE %g, seed to seed %g" % (rmse, rmse_floor)
    assert rel_mse <= 1.5 * floor, "relMSE %g, seed to seed %g" % (rel_mse, floor)


def test_device_tonemap_matches_host(pkg, ctx):
    """pt_film_tonemap16 against the host tone map (reference src/image.cpp:12-33) on the same film: 16-bit samples
    equal, except where the device's expf/powf differ from libm's in the last place -- then by one step at most."""
    W, H, spp = 96, 64, 16
    scene = S.knot_scene(pkg, W, H)
    ctx.upload(scene)
    ctx.film_bind(W, H)
    ctx.render(0, spp, spp, seed=4)
    film = ctx.film_download()
    dev = ctx.film_tonemap16(2.0, 2.2).astype(np.int64)
    host = pkg.tonemap16(film, 2.0, 2.2).astype(np.int64).reshape(dev.shape)
    diff = np.abs(dev - host)
    assert diff.max() <= 1, "device and host tone map differ by %d steps" % diff.max()
    assert (diff == 0).mean() >= 0.98, "only %.4f of the samples are equal" % (diff == 0).mean()
    assert dev.max() > 1000 and dev.min() >= 0


def read_pfm(path):
    with open(path, "rb") as fh:
        assert fh.readline().strip() == b"PF"
        w, h = map(int, fh.readline().split())
        assert float(fh.readline()) < 0          # little endian
        return np.frombuffer(fh.read(), "<f4").reshape(h, w, 3)[::-1]


def test_film_reduce_across_contexts(pkg, ctx):
    """Two contexts render half of the samples each (as two GPUs driven by one process would); pt_film_reduce adds the
    second film to the first: bit-equal to adding the downloaded films, and within float addition order of one call."""
    W, H, spp = 96, 64, 8
    scene = S.knot_scene(pkg, W, H)
    other = pkg.Context(0)
    try:
        for c in (ctx, other):
            c.upload(scene)
            c.film_bind(W, H)
        ctx.render(0, spp // 2, spp, seed=6)
        other.render(spp // 2, spp - spp // 2, spp, seed=6)
        a, b = ctx.film_download(), other.film_download()
        pkg.film_reduce([ctx, other], root=0)
        total = ctx.film_download()
        assert np.array_equal(total, a + b)
        assert np.array_equal(other.film_download(), b)
        ctx.film_clear()
        ctx.render(0, spp, spp, seed=6)
        whole = ctx.film_download()
        assert np.allclose(total, whole, rtol=1e-5, atol=1e-7)
        with pytest.raises(pkg.PtError):
            pkg.film_reduce([ctx, ctx])
        other.film_bind(W + 2, H)
        with pytest.raises(pkg.PtError):
            pkg.film_reduce([ctx, other])
    finally:
        other.close()


def test_film_reduce_binds_rccl(pkg, ctx, monkeypatch):
    """Contexts on distinct GPUs are summed by one ncclReduce; this box has one GPU, so the binding (library lookup, symbols,
    communicator, the call on the context's stream) is exercised with a one-rank reduce, which must leave the film as is."""
    W, H, spp = 96, 64, 2
    ctx.upload(S.knot_scene(pkg, W, H))
    ctx.film_bind(W, H)
    ctx.render(0, spp, spp, seed=6)
    before = ctx.film_download()
    monkeypatch.setenv("PT_FILM_REDUCE_RCCL", "1")
    pkg.film_reduce([ctx], root=0)
    pkg.film_reduce([ctx], root=0)              # the communicator is reused
    assert np.array_equal(ctx.film_download(), before)
    monkeypatch.delenv("PT_FILM_REDUCE_RCCL")
    pkg.film_reduce([ctx], root=0)
    assert np.array_equal(ctx.film_download(), before)


def test_command_line_matches_library(pkg, ctx, tmp_path):
    """The CLI (same arguments as the reference's, src/main.cu:16-26) writes the PNG and, with PT_PFM=1, the linear
    film: bit-equal to the film the library renders for the same scene, size, samples and seed; with the samples
    split over two contexts (PT_DEVICES=0,0) equal up to float addition order."""
    import subprocess
    exe = os.path.join(pkg.PKG_DIR, "pathtracer")
    scene_path = S.scene_file(pkg, "balls")
    W, H, spp = 160, 90, 8
    out = str(tmp_path / "img")
    env = dict(os.environ, PT_PFM="1", PT_SEED="5")
    r = subprocess.run([exe, scene_path, str(W), str(H), str(spp), out], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "Rendered image in" in r.stdout and "Saved image in" in r.stdout
    assert open(out + ".png", "rb").read(8) == b"\x89PNG\r\n\x1a\n"
    cli = read_pfm(out + ".pfm")
    hs = pkg.HostScene(scene_path, W, H)
    ctx.upload(hs)
    ctx.film_bind(W, H)
    ctx.render(0, spp, spp, seed=5)
    assert np.array_equal(cli, ctx.film_download())
    out2 = str(tmp_path / "img2")
    r = subprocess.run([exe, scene_path, str(W), str(H), str(spp), out2], capture_output=True, text=True,
                       env=dict(env, PT_DEVICES="0,0"), timeout=120)
    assert r.returncode == 0, r.stderr
    assert '"gpus": 2' in r.stdout
    assert np.allclose(read_pfm(out2 + ".pfm"), cli, rtol=1e-5, atol=1e-7)


def test_bench_starts_its_own_ranks(pkg):
    """`python bench.py --gpus 2` with no launcher around it starts its two ranks itself and relays rank 0's line.  Rehearsal
    form (PT_BENCH_REHEARSAL=1): both ranks on this one GPU, gloo for the reduce; the two ranks' shares of the samples sum
    to the rays of a one-rank run of the same frame."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    args = ["--scene", "bunny", "--width", "256", "--height", "144", "--spp", "8", "--steps", "1", "--warmup", "0",
            "--no-cpu-baseline"]
    lines = []
    for gpus in (2, 1):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus)] + args,
                           env=dict(env, PT_BENCH_REHEARSAL="1"), cwd=root, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(out) == 1
        lines.append(json.loads(out[0]))
    two, one = lines
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["config"]["rays_per_frame"] == one["config"]["rays_per_frame"]
    assert abs(two["film_check"]["mean"] - one["film_check"]["mean"]) <= 1e-5 * one["film_check"]["mean"]


def test_film_reduce_through_torch_rccl_backend(pkg, tmp_path):
    """The collective of the one-process-per-GPU form is torch.distributed's reduce on the film tensor the library renders
    into, backend "nccl" (RCCL).  One GPU allows one rank: a one-rank group, the reduce in place.  Checks what a single GPU
    can: the process group comes up beside the library (both bind the HIP runtime PyTorch brought), the reduce runs on the
    tensor the library wrote through its own streams, and the film is the one rendered without it."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "one_rank.py"
    script.write_text("""
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import __graft_entry__ as G
import scenes as S
pkg = G.load_package()
import cuda_path_tracer_amd.distributed as D
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%%d" %% (29600 + os.getpid() %% 300), rank=0, world_size=1)
W, H, spp = 96, 64, 6
ctx = pkg.Context(0)
ctx.upload(S.knot_scene(pkg, W, H))
film = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
ctx.film_bind(W, H, film.data_ptr())
torch.cuda.synchronize()
D.render_and_reduce(lambda b, c, t, s: ctx.render(b, c, t, s), film, spp, seed=3)
dist.reduce(film, dst=0, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
np.save(sys.argv[1], film.cpu().numpy())
ctx.close()
dist.destroy_process_group()
""" % (root, root))
    out = str(tmp_path / "film.npy")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(script), out], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    W, H, spp = 96, 64, 6
    scene = S.knot_scene(pkg, W, H)
    ofilm, _ = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=3)
    assert_same_film(np.load(out), ofilm, "film after a one-rank RCCL reduce")


def test_render_is_partition_invariant(pkg, ctx):
    """Samples split over two calls (as two GPUs would) sum to the single-call film up to float addition order."""
    W, H, spp = 96, 64, 8
    scene = S.knot_scene(pkg, W, H)
    ctx.upload(scene)
    ctx.film_bind(W, H)
    ctx.render(0, spp, spp, seed=3)
    full = ctx.film_download()
    ctx.film_clear()
    parts = []
    for r in range(2):
        b, c = pkg.spp_partition(spp, 2, r)
        ctx.film_clear()
        ctx.render(b, c, spp, seed=3)
        parts.append(ctx.film_download())
    summed = parts[0] + parts[1]
    assert np.allclose(summed, full, rtol=1e-5, atol=1e-6)


def test_render_small_pool_many_iterations(pkg, ctx):
    """A pool far smaller than the sample count forces many wavefront iterations and queue wrap-arounds."""
    W, H, spp = 64, 48, 4
    scene = S.knot_scene(pkg, W, H)
    ctx.upload(scene)
    ctx.film_bind(W, H)
    ctx.set_option(pkg.OPT_RENDER_MODE, 1)            # the pool belongs to the wavefront loop
    try:
        ctx.set_option(pkg.OPT_POOL_SIZE, 1 << 22)
        ctx.render(0, spp, spp, seed=9)
        big = ctx.film_download()
        ctx.film_clear()
        ctx.set_option(pkg.OPT_POOL_SIZE, 1024)
        st = ctx.render(0, spp, spp, seed=9)
        small = ctx.film_download()
    finally:
        ctx.set_option(pkg.OPT_POOL_SIZE, 1 << 22)
        ctx.set_option(pkg.OPT_RENDER_MODE, 0)
    assert st.iterations >= W * H * spp // 1024
    assert st.paths == W * H * spp
    assert np.array_equal(small, big), "the film depends on the pool size"


def test_render_film_size_that_does_not_tile(pkg, ctx):
    """50x30 is not a multiple of the 8x4 pixel tiles: paths are enumerated row by row instead, the per-launch index
    arithmetic (one division per thread, then additions with carry) wraps many times with a 1024-slot pool, and the film
    must still be the oracle's and must not depend on the pool."""
    W, H, spp = 50, 30, 5
    scene = S.knot_scene(pkg, W, H)
    ctx.upload(scene)
    ctx.film_bind(W, H)
    films = []
    ctx.set_option(pkg.OPT_RENDER_MODE, 1)
    try:
        for pool in (1 << 22, 1024, 4096 + 512):
            ctx.set_option(pkg.OPT_POOL_SIZE, pool)
            ctx.film_clear()
            st = ctx.render(0, spp, spp, seed=13)
            assert st.paths == W * H * spp
            films.append(ctx.film_download())
    finally:
        ctx.set_option(pkg.OPT_POOL_SIZE, 1 << 22)
        ctx.set_option(pkg.OPT_RENDER_MODE, 0)
    assert np.array_equal(films[0], films[1]) and np.array_equal(films[0], films[2])
    ofilm, ost = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=13)
    assert st.rays == ost.rays
    assert_same_film(films[0], ofilm, "small pool")


def test_packed_environment_texels_change_nothing(pkg):
    """A map decoded from a Radiance .hdr file is kept as 8-byte texels (shared-exponent rgb + pdf); the film is bit
    for bit the one rendered from the 16-byte float texels.  A map of arbitrary floats is not packed."""
    W, H, spp = 96, 64, 4
    scene = pkg.HostScene(S.scene_file(pkg, "balls"), W, H)
    films = []
    for pack in (1, 0):
        c = pkg.Context(0)
        c.set_option(pkg.OPT_ENV_PACK, pack)
        c.upload(scene)
        assert c.env_is_packed == bool(pack)
        c.film_bind(W, H)
        c.render(0, spp, spp, seed=11)
        films.append(c.film_download())
        c.close()
    assert np.array_equal(films[0], films[1])
    c = pkg.Context(0)
    c.upload(S.knot_scene(pkg, W, H))
    assert not c.env_is_packed
    c.close()


@pytest.mark.parametrize("size", [(1, 1), (2, 1), (5, 3), (7, 4), (13, 9), (64, 33)])
def test_environment_maps_of_odd_sizes(pkg, size):
    """Environment maps whose sides are not multiples of the 4x4 texel tiles of the packed layout, down to a single texel:
    bilinear footprints wrap around both edges, the tables have one column.  Packed and unpacked texels give the same
    film, bit-equal to the oracle's."""
    ew, eh = size
    W, H, spp = 64, 40, 4
    rng = np.random.default_rng(ew * 100 + eh)
    tex = np.zeros((eh, ew, 4), dtype=np.float32)
    # values a Radiance pixel can hold: three 8-bit mantissas times one power of two per texel
    mant = rng.integers(0, 256, size=(eh, ew, 3)).astype(np.float32)
    expo = rng.integers(-6, 3, size=(eh, ew, 1)).astype(np.float32)
    tex[..., :3] = mant * np.exp2(expo)
    tex[0, 0, :3] = [200.0, 180.0, 90.0]               # one bright texel for the light sampling to find
    tex[..., 3] = 1.0
    marg, cond = pkg.env_build(tex)
    cam = pkg.camera_make((0, 1, 5), (0, 0.5, 0), 35.0, W / H, 0.1, 4.75)
    planes = [pkg.make_plane((0, 0, 0), (0, 1, 0), (0.325, 0.3, 0.35, 0.0), pkg.DIFFUSE)]
    spheres = [pkg.make_sphere((-0.7, 0.5, 0), 0.5, (0.7, 0.7, 0.7, 0.0), pkg.DIFFUSE),
               pkg.make_sphere((0.7, 0.5, 0), 0.5, (0.15, 0.25, 0.4, 0.05), pkg.GLOSSY)]
    scene = pkg.ManualScene(cam, tex, marg, cond, planes=planes, spheres=spheres)
    ofilm, ost = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=23)
    # (a one-row map sends every light sample straight down, where both pdfs of src/render.cu:145-151 vanish on the floor:
    # the reference's 0/0; the NaN pixels must be the same pixels on both sides)
    assert np.isfinite(ofilm).all() or eh == 1
    for pack in (1, 0):
        c = pkg.Context(0)
        c.set_option(pkg.OPT_ENV_PACK, pack)
        c.upload(scene)
        assert c.env_is_packed == bool(pack)
        c.film_bind(W, H)
        st = c.render(0, spp, spp, seed=23)
        film = c.film_download()
        c.close()
        assert st.rays == ost.rays
        assert_same_film(film, ofilm, "environment %dx%d, packed %d" % (ew, eh, pack))


def test_rays_settled_where_they_are_made_are_counted(pkg, ctx):
    """Rays whose closest hit follows from the plane/sphere tests and the BVH root's child boxes never reach an extend
    launch but still count as rays: with no mesh that is every ray, with one it is a part, and the totals stay the
    oracle's (ray counts within the flipped-path rate, node visits equal on hit-for-hit identical batches)."""
    W, H, spp = 96, 64, 4
    scene = S.spheres_scene(pkg, W, H)
    ctx.upload(scene)
    ctx.film_bind(W, H)
    st = ctx.render(0, spp, spp, seed=3)
    _, ost = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=3)
    assert st.rays_escaped == st.rays and st.extend_launches == 0
    assert abs(int(st.rays) - int(ost.rays)) <= 0.002 * ost.rays
    scene = S.knot_scene(pkg, W, H)
    ctx.upload(scene)
    ctx.film_bind(W, H)
    st = ctx.render(0, spp, spp, seed=3)
    _, ost = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=3)
    assert 0 < st.rays_escaped < st.rays
    assert abs(int(st.rays) - int(ost.rays)) <= 0.002 * ost.rays


def test_film_does_not_depend_on_the_stream_count(pkg, ctx):
    """pt_render runs up to four wavefront loops side by side, each on a share of the samples; which loop renders a
    sample must not show in the film."""
    W, H, spp = 96, 64, 7
    scene = S.knot_scene(pkg, W, H)
    ctx.upload(scene)
    ctx.film_bind(W, H)
    films = []
    ctx.set_option(pkg.OPT_RENDER_MODE, 1)
    try:
        for lanes in (1, 2, 3, 4):
            ctx.set_option(pkg.OPT_STREAMS, lanes)
            ctx.film_clear()
            st = ctx.render(0, spp, spp, seed=21)
            assert st.paths == W * H * spp
            films.append(ctx.film_download())
    finally:
        ctx.set_option(pkg.OPT_STREAMS, 3)
        ctx.set_option(pkg.OPT_RENDER_MODE, 0)
    for f in films[1:]:
        assert np.array_equal(f, films[0])
    with pytest.raises(pkg.PtError):
        ctx.set_option(pkg.OPT_STREAMS, 5)


def test_render_in_several_slabs(pkg, ctx):
    """A contribution slab that holds fewer samples than the call renders: the call proceeds slab after slab and the
    film is the same (per-pixel sums are taken per slab, so equal up to float addition order)."""
    W, H, spp = 64, 48, 6
    scene = S.knot_scene(pkg, W, H)
    ctx.upload(scene)
    ctx.film_bind(W, H)
    ctx.render(0, spp, spp, seed=9)
    one = ctx.film_download()
    again = None
    ctx.film_clear()
    ctx.render(0, spp, spp, seed=9)
    again = ctx.film_download()
    assert np.array_equal(one, again), "the film is not reproducible bit for bit"
    ctx.film_clear()
    with pytest.raises(pkg.PtError):
        ctx.set_option(pkg.OPT_SLAB_BYTES, 16)
    c2 = pkg.Context(0)
    c2.upload(scene)
    c2.film_bind(W, H)
    c2.set_option(pkg.OPT_SLAB_BYTES, 1 << 20)          # 1 MiB = 21 samples of 64x48 ... use a larger film below
    c2.close()
    W2, H2 = 256, 192                                    # 768 KiB per sample: a 1 MiB slab holds one sample
    scene2 = S.knot_scene(pkg, W2, H2)
    c3 = pkg.Context(0)
    c3.upload(scene2)
    c3.film_bind(W2, H2)
    c3.render(0, 3, 3, seed=4)
    ref = c3.film_download()
    c3.film_clear()
    c3.set_option(pkg.OPT_SLAB_BYTES, 1 << 20)
    st = c3.render(0, 3, 3, seed=4)
    got = c3.film_download()
    c3.close()
    assert st.paths == W2 * H2 * 3
    assert np.allclose(got, ref, rtol=1e-6, atol=1e-7)


def test_traversal_counters_match_oracle(pkg, ctx):
    """Inner nodes visited, triangles tested and leaves entered are per-ray deterministic: the sums must be equal."""
    W, H = 48, 32
    scene = S.parity_cases(pkg)["knot"]
    g = np.load(os.path.join(GOLDEN, "rays_knot.npz"))
    _, _, _, ost = O.intersect(S.as_oracle_desc(scene), g["org"], g["dir"])
    import torch

    ctx.upload(scene)
    ctx.set_option(pkg.OPT_COUNT_TRAVERSAL, 1)
    try:
        n = g["org"].shape[0]
        o4 = torch.zeros((n, 4), dtype=torch.float32)
        d4 = torch.zeros((n, 4), dtype=torch.float32)
        o4[:, :3] = torch.from_numpy(g["org"])
        d4[:, :3] = torch.from_numpy(g["dir"])
        o4, d4 = o4.cuda(), d4.cuda()
        hit = torch.empty((n, 4), dtype=torch.float32, device="cuda")
        t = torch.empty((n,), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        st = ctx.intersect_device(n, o4.data_ptr(), d4.data_ptr(), hit.data_ptr(), t.data_ptr())
    finally:
        ctx.set_option(pkg.OPT_COUNT_TRAVERSAL, 0)
    assert (st.nodes, st.tris, st.leaves) == (ost.nodes, ost.tris, ost.leaves)
    assert np.array_equal(t.cpu().numpy(), g["t"])
    assert np.array_equal(hit.cpu().numpy()[:, 3].view(np.uint32), g["prim"])



def _whole_path_scenes(pkg, W, H):
    return {
        "knot": lambda: S.knot_scene(pkg, W, H),                                   # glossy mesh, all four materials around it
        "balls": lambda: S.spheres_scene(pkg, W, H),                               # no tree: every ray is settled where it is made
        "two_meshes": lambda: pkg.HostScene(S.scene_file(pkg, "two_meshes"), W, H),   # per-mesh materials
        "suzanne": lambda: pkg.HostScene(S.scene_file(pkg, "suzanne"), W, H),      # pinhole camera
    }


@pytest.mark.parametrize("name", ["knot", "balls", "two_meshes", "suzanne"])
def test_whole_path_kernel_matches_oracle(pkg, ctx, name):
    """PT_OPT_RENDER_MODE 2 (every path traced from its first ray to its end by one lane of a persistent kernel): film, ray
    count, settled-ray count and traversal counters equal to the oracle's and to the wavefront loop's, also with a traversal
    stack that overflows into global memory and with a service threshold that keeps waves nearly empty or nearly full."""
    W, H, spp = 96, 64, 6
    scene = _whole_path_scenes(pkg, W, H)[name]()
    ctx.upload(scene)
    ctx.film_bind(W, H)
    ofilm, ost = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=21)
    ctx.set_option(pkg.OPT_COUNT_TRAVERSAL, 1)
    try:
        ctx.set_option(pkg.OPT_RENDER_MODE, 1)
        wst = ctx.render(0, spp, spp, seed=21)
        assert ctx.render_mode_used == 1
        assert_same_film(ctx.film_download(), ofilm, name + " wavefront")
        ctx.set_option(pkg.OPT_RENDER_MODE, 2)
        for refill, lds in ((44, 32), (1, 32), (64, 32), (44, 3)):
            ctx.set_option(pkg.OPT_TUNE_PATH_REFILL, refill)
            ctx.set_option(pkg.OPT_TUNE_PATH_LDS_SLOTS, lds)
            ctx.film_clear()
            st = ctx.render(0, spp, spp, seed=21)
            what = "%s threshold %d, %d LDS slots" % (name, refill, lds)
            assert_same_film(ctx.film_download(), ofilm, what)
            assert ctx.render_mode_used == 2
            assert st.paths == W * H * spp and st.rays == ost.rays == wst.rays, what
            assert st.rays_escaped == wst.rays_escaped, what
            assert (st.nodes, st.tris, st.leaves) == (ost.nodes, ost.tris, ost.leaves) == (wst.nodes, wst.tris, wst.leaves), what
    finally:
        ctx.set_option(pkg.OPT_RENDER_MODE, 0)
        ctx.set_option(pkg.OPT_TUNE_PATH_REFILL, 52)
        ctx.set_option(pkg.OPT_TUNE_PATH_LDS_SLOTS, 28)
        ctx.set_option(pkg.OPT_COUNT_TRAVERSAL, 0)


def test_render_mode_is_chosen_from_the_first_sample(pkg, ctx):
    """PT_OPT_RENDER_MODE 0 (the default): the first render after an upload traces its first sample with the whole-path kernel
    and picks the mode for the rest from the share of that sample's rays that were settled where they were made.  A mesh that
    fills the frame gets mode 2, a scene of spheres mode 1; the film is the oracle's either way, counters included, also when
    the call runs slab after slab (the pilot sample is part of the first slab) and on the calls after the choice was made."""
    W, H, spp = 96, 64, 6
    cases = {"city": (lambda: pkg.HostScene(S.scene_file(pkg, "city"), W, H), 2),
             "balls": (lambda: S.spheres_scene(pkg, W, H), 1)}
    for name, (make, want_mode) in cases.items():
        scene = make()
        ofilm, ost = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=31)
        ctx.upload(scene)                                    # forgets an earlier choice
        ctx.film_bind(W, H)
        ctx.set_option(pkg.OPT_COUNT_TRAVERSAL, 1)
        try:
            for call in range(2):                            # the first call chooses, the second finds the choice made
                ctx.film_clear()
                st = ctx.render(0, spp, spp, seed=31)
                assert ctx.render_mode_used == want_mode, name
                assert_same_film(ctx.film_download(), ofilm, "%s, call %d" % (name, call))
                assert st.paths == W * H * spp and st.rays == ost.rays
                assert (st.nodes, st.tris, st.leaves) == (ost.nodes, ost.tris, ost.leaves)
            ctx.upload(scene)
            ctx.set_option(pkg.OPT_SLAB_BYTES, 1 << 20)
            ctx.film_bind(W * 4, H * 4)                      # 1.5 MB of slab per sample, 1 MiB allowed: a slab per sample
            st = ctx.render(0, 3, 3, seed=31)
            # (the camera depends on the film's aspect ratio only, so the same scene serves the larger film)
            big, _ = O.render(S.as_oracle_desc(scene), W * 4, H * 4, 0, 3, 3, seed=31)
            assert st.paths == W * H * 16 * 3
            assert np.allclose(ctx.film_download(), big, rtol=1e-5, atol=1e-7)
        finally:
            ctx.set_option(pkg.OPT_COUNT_TRAVERSAL, 0)
            ctx.set_option(pkg.OPT_SLAB_BYTES, 32 << 30)
    # a single-sample call is the pilot alone
    scene = S.spheres_scene(pkg, W, H)
    ctx.upload(scene)
    ctx.film_bind(W, H)
    st = ctx.render(2, 1, 4, seed=3)
    o, ost = O.render(S.as_oracle_desc(scene), W, H, 2, 1, 4, seed=3)
    assert st.rays == ost.rays
    assert_same_film(ctx.film_download(), o, "one sample")


def test_whole_path_kernel_slabs_partition_and_odd_sizes(pkg, ctx):
    """Render mode 2 on a film that does not tile into 8x4 pixel blocks and whose pixel count is not a multiple of the chunk
    of camera paths a wave claims: one slab, several slabs and a split of the samples over two calls give the oracle's film
    (the split: up to float addition order)."""
    W, H, spp = 250, 150, 7
    scene = S.knot_scene(pkg, W, H)
    ctx.upload(scene)
    ctx.film_bind(W, H)
    ofilm, ost = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=5)
    ctx.set_option(pkg.OPT_RENDER_MODE, 2)
    try:
        st = ctx.render(0, spp, spp, seed=5)
        assert st.rays == ost.rays
        assert_same_film(ctx.film_download(), ofilm, "one slab")
        ctx.set_option(pkg.OPT_SLAB_BYTES, 1 << 20)              # 1 MiB: one sample of this film (600 KB) per slab
        ctx.film_clear()
        st = ctx.render(0, spp, spp, seed=5)
        assert st.rays == ost.rays and st.iterations == spp
        assert_same_film(ctx.film_download(), ofilm, "several slabs")
        ctx.film_clear()
        ctx.render(0, 3, spp, seed=5)
        ctx.render(3, spp - 3, spp, seed=5)
        assert np.allclose(ctx.film_download(), ofilm, rtol=1e-5, atol=1e-7)
    finally:
        ctx.set_option(pkg.OPT_RENDER_MODE, 0)
        ctx.set_option(pkg.OPT_SLAB_BYTES, 32 << 30)


def test_scene_without_any_primitive(pkg, ctx):
    """A scene file may hold a camera and an environment and nothing else: every camera ray leaves at once (one ray per path,
    all settled where they are made), queries return misses, and the film is the oracle's in every render mode."""
    cam = pkg.camera_make((0, 1, 5), (0, 0.5, 0), 35.0, 1.5, 0.1, 4.75)
    tex, marg, cond = S.small_env(pkg)
    scene = pkg.ManualScene(cam, tex, marg, cond)
    W, H, spp = 48, 32, 8
    want, ost = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=4)
    assert ost.rays == ost.paths == W * H * spp
    ctx.upload(scene)
    org = np.zeros((65, 3), np.float32)
    dir_ = np.tile(np.array([[0.0, 0.6, -0.8]], np.float32), (65, 1))
    t, prim = ctx.intersect(org, dir_)[:2]
    assert (prim == 0xFFFFFFFF).all() and (t == np.float32(3.402823466e38)).all()
    ctx.film_bind(W, H)
    try:
        for mode in (1, 2, 0):
            ctx.set_option(pkg.OPT_RENDER_MODE, mode)
            ctx.film_clear()
            st = ctx.render(0, spp, spp, seed=4)
            assert_same_film(ctx.film_download(), want, "no primitives, mode %d" % mode)
            assert st.rays == st.paths == W * H * spp and st.rays_escaped == st.rays
    finally:
        ctx.set_option(pkg.OPT_RENDER_MODE, 0)


def test_mesh_with_degenerate_triangles_matches_oracle(pkg, ctx):
    """Triangles without area get the identity transform from the reference's builder (tests/scenes.py::degenerate_mesh): a
    phantom unit triangle at the origin behind each one's leaf box.  Rays aimed through that phantom at the degenerate
    triangles find it where the reference's arithmetic says -- the device's hits are the oracle's, phantom hits included --
    and a film of the mesh is the oracle's."""
    tris, idx = S.degenerate_mesh()
    bvh = pkg.bvh_build(tris)
    cam = pkg.camera_make((0.2, 1.2, 3.5), (-0.1, 0.8, 0), 40.0, 1.5, 0.0, 3.5)
    tex, marg, cond = S.small_env(pkg)
    scene = pkg.ManualScene(cam, tex, marg, cond, bvh=bvh, mesh_material=(0.6, 0.6, 0.6, 0.0),
                            mesh_material_type=pkg.DIFFUSE)
    rng = np.random.default_rng(1)
    n = 20000
    target = tris[rng.choice(idx, n), 0].astype(np.float64)
    uv = rng.uniform(0, 1, (n, 2))
    uv = np.where((uv.sum(axis=1) > 1)[:, None], 1 - uv, uv)
    through = np.concatenate([uv, np.zeros((n, 1))], axis=1)          # a point of the phantom triangle
    d = target - through
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    org = (through - 0.5 * d).astype(np.float32)
    d = d.astype(np.float32)
    desc = S.as_oracle_desc(scene)
    ot, oprim, onrm, _ = O.intersect(desc, org, d, want_normals=True)
    assert np.isin(oprim[oprim != 0xFFFFFFFF], idx).sum() > 100       # the phantoms are found
    ctx.upload(scene)
    t, prim, nrm = ctx.intersect(org, d, want_normals=True)[:3]
    assert np.array_equal(prim, oprim) and np.array_equal(t, ot)
    hit = oprim != 0xFFFFFFFF
    assert np.array_equal(nrm[hit].view(np.uint32), onrm[hit].view(np.uint32))
    W, H, spp = 96, 64, 8
    want, ost = O.render(desc, W, H, 0, spp, spp, seed=6)
    ctx.film_bind(W, H)
    try:
        for mode in (1, 2):
            ctx.set_option(pkg.OPT_RENDER_MODE, mode)
            ctx.film_clear()
            st = ctx.render(0, spp, spp, seed=6)
            assert_same_film(ctx.film_download(), want, "degenerate triangles, mode %d" % mode)
            assert st.rays == ost.rays
    finally:
        ctx.set_option(pkg.OPT_RENDER_MODE, 0)


def test_hostile_rays_give_the_oracles_answers(pkg, ctx):
    """Rays no renderer makes but a caller of the query entry point may pass: zero, axis-parallel, infinite and NaN
    directions, origins at infinity, NaN or 1e30 away, not-normalised and denormal directions.  The traversal ends (it
    visits a finite tree) and returns what the reference's arithmetic returns for them: the oracle's t, id and normal, bit
    for bit, NaN for NaN."""
    scene = S.parity_cases(pkg)["knot"]
    base_o, base_d = O.primary_rays(S.as_oracle_desc(scene), 24, 16, sample=0, seed=8)
    rng = np.random.default_rng(8)
    specials = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e30, -1e30, 1e-42, 3e38], np.float32)
    n = 4096
    org = base_o[rng.integers(0, len(base_o), n)].copy()
    d = base_d[rng.integers(0, len(base_d), n)].copy()
    for arr in (org, d):
        which = rng.random((n, 3)) < 0.35
        arr[which] = specials[rng.integers(0, len(specials), which.sum())]
    d[:8] = 0.0                                              # the zero direction, from eight origins
    ctx.upload(scene)
    ot, oprim, onrm, _ = O.intersect(S.as_oracle_desc(scene), org, d, want_normals=True)
    for mode in (2, 1):
        ctx.set_option(pkg.OPT_EXTEND_MODE, mode)
        t, prim, nrm = ctx.intersect(org, d, want_normals=True)[:3]
        assert np.array_equal(prim, oprim), mode
        assert np.array_equal(t.view(np.uint32), ot.view(np.uint32)) or \
            ((t.view(np.uint32) == ot.view(np.uint32)) | (np.isnan(t) & np.isnan(ot))).all(), mode
        hit = oprim != 0xFFFFFFFF
        same = (nrm[hit].view(np.uint32) == onrm[hit].view(np.uint32)) | (np.isnan(nrm[hit]) & np.isnan(onrm[hit]))
        assert same.all(), mode
    ctx.set_option(pkg.OPT_EXTEND_MODE, 2)
    assert (oprim != 0xFFFFFFFF).sum() > 50                  # some of them do hit something


def test_error_paths(pkg):
    c = pkg.Context(0)
    with pytest.raises(pkg.PtError):
        c.intersect(np.zeros((1, 3), np.float32), np.ones((1, 3), np.float32))       # no scene
    scene = S.spheres_scene(pkg, 48, 32)
    c.upload(scene)
    with pytest.raises(pkg.PtError):
        c.render(0, 1, 1)                                                             # no film
    c.film_bind(48, 32)
    with pytest.raises(pkg.PtError):
        c.render(3, 2, 4)                                                             # range beyond spp_total
    with pytest.raises(pkg.PtError):
        c.set_option(pkg.OPT_POOL_SIZE, 8)
    with pytest.raises(pkg.PtError):
        pkg.Context(99)
    # a scene that is refused leaves the uploaded one in place: unknown material type, a count without its array
    bad = S.spheres_scene(pkg, 48, 32)
    C.cast(bad.desc.spheres, C.POINTER(pkg.Sphere))[1].material_type = 0x40
    with pytest.raises(pkg.PtError, match="invalid material type"):
        c.upload(bad)
    d = pkg.SceneDesc.from_buffer_copy(scene.desc)
    d.planes = None
    with pytest.raises(pkg.PtError, match="without primitive array"):
        c.upload(d)
    # an empty sample range is a no-op; a single sample uses one stream
    st = c.render(2, 0, 4)
    assert st.paths == 0 and st.rays == 0 and not c.film_download().any()
    st = c.render(2, 1, 4)
    assert st.paths == 48 * 32
    one = c.film_download()
    o, _ = O.render(S.as_oracle_desc(scene), 48, 32, 2, 1, 4, seed=0)
    assert_same_film(one, o, "sample 2 of 4")
    c.close()


def test_upload_rejects_malformed_bvh_buffers(pkg):
    """pt_scene_upload validates what it re-lays (in parallel passes for large meshes): a child index outside the node
    array, a leaf reference past the triangle buffer, a leaf run without terminator and a cycle are refused with an
    error instead of producing a tree the traversal could run off."""
    tris = S.knot_mesh(pkg)
    rng = np.random.default_rng(3)
    big = np.concatenate([tris + rng.uniform(-2, 2, size=(1, 1, 3)).astype(np.float32) for _ in range(4)])
    good = pkg.bvh_build(big, threads=4)                  # 36,864 triangles: enough nodes for several upload threads
    cam = pkg.camera_make((0, 1, 6), (0, 0, 0), 40.0, 1.5, 0.0, 6.0)
    tex, marg, cond = S.small_env(pkg)
    c = pkg.Context(0)

    def upload(bvh):
        c.upload(pkg.ManualScene(cam, tex, marg, cond, bvh=bvh))

    upload(good)                                          # the untouched buffers are fine
    n_nodes = good["nodes"].shape[0]

    def broken(edit):
        b = dict(good)
        b["nodes"] = good["nodes"].copy()
        b["tris"] = good["tris"].copy()
        edit(b)
        return b

    def child_out_of_range(b):
        b["nodes"].view(np.int32)[n_nodes // 2, 12] = n_nodes + 5

    def leaf_past_the_buffer(b):
        b["nodes"].view(np.int32)[n_nodes - 1, 13] = ~np.int32(good["num_tri_float4"] + 7)

    def run_without_terminator(b):
        t = b["tris"].view(np.uint32)
        t[t[:, 0] == 0x80000000, 0] = 0x3F800000          # no float4 ends a run any more

    def cycle(b):
        b["nodes"].view(np.int32)[n_nodes - 1, 12] = 1    # a deep node points back up: not a tree

    ray = (np.array([[0, 1, 6]], np.float32), np.array([[0, -0.16, -1]], np.float32))
    t0, prim0 = c.intersect(*ray)[:2]
    for edit in (child_out_of_range, leaf_past_the_buffer, run_without_terminator, cycle):
        with pytest.raises(pkg.PtError):
            upload(broken(edit))
        # a refused scene leaves the one in place untouched
        t, prim = c.intersect(*ray)[:2]
        assert np.array_equal(t, t0) and np.array_equal(prim, prim0)
    upload(good)                                          # and the context still takes a good one afterwards
    t, prim = c.intersect(*ray)[:2]
    assert np.array_equal(t, t0) and np.array_equal(prim, prim0)
    c.close()


def test_full_size_frame_properties(pkg, ctx):
    """At the benchmark's resolution (1920x1080) the oracle is too slow to run in full, so check size-independent
    properties: every path is started exactly once, splitting the samples does not change the film, the film is
    finite and non-negative, and the ray count equals the sum over both halves."""
    W, H, spp = 1920, 1080, 2
    scene = S.knot_scene(pkg, W, H)
    ctx.upload(scene)
    ctx.film_bind(W, H)
    st = ctx.render(0, spp, spp, seed=1)
    full = ctx.film_download()
    assert st.paths == W * H * spp
    assert np.isfinite(full).all() and (full >= 0).all()
    acc = np.zeros_like(full)
    rays = 0
    for r in range(2):
        b, c = pkg.spp_partition(spp, 2, r)
        ctx.film_clear()
        s2 = ctx.render(b, c, spp, seed=1)
        rays += s2.rays
        acc += ctx.film_download()
    assert rays == st.rays
    assert np.allclose(acc, full, rtol=1e-5, atol=1e-6)
    # and against the oracle: the same frame rendered by the CPU, bit for bit
    ofilm, _ = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=1, nthreads=16)
    assert_same_film(full, ofilm, "1920x1080")


def test_more_than_2_to_32_paths_in_one_call(pkg, ctx):
    """The largest benchmark configuration has 8.5 G camera paths (3840x2160x1024): path and ray counters are 64-bit, and a
    call whose samples exceed the slab budget runs slab after slab.  Here 3840x2160x520 = 4.31 G paths (> 2^32) of the
    spheres-only scene in one call: every path started exactly once, and the film equal to a 40-sample render of the same
    scene within the noise of 40 samples."""
    W, H, spp = 3840, 2160, 520
    scene = S.spheres_scene(pkg, W, H)
    ctx.upload(scene)
    ctx.film_bind(W, H)
    st = ctx.render(0, spp, spp, seed=9)
    assert st.paths == W * H * spp and st.paths > 2 ** 32
    assert st.rays > st.paths
    big = ctx.film_download()
    assert np.isfinite(big).all() and (big >= 0).all()
    ctx.film_clear()
    ctx.render(0, 40, 40, seed=9)
    small = ctx.film_download()
    # same estimator, fewer samples: the means agree closely, single pixels within the noise of 40 samples
    assert abs(big.mean() - small.mean()) <= 0.01 * small.mean()
    rel_mse, _ = film_metrics(small, big)
    assert rel_mse <= 0.2, rel_mse


def test_hundred_million_triangle_scene_matches_oracle(pkg, ctx):
    """The largest tree of the benchmark configurations at its full size: 116 copies of the 868,352-triangle stand-in in ONE BVH
    (100.7 M triangles, 9.6 GB of records on the device, glossy and glass meshes alternating, glass spheres between
    them).  The tree is deeper than the traversal stack's LDS column, so part of every deep ray's stack lives in global
    memory.  Hits bit-equal to the oracle on primary + bounced + axis-parallel rays; a small film bit-equal too."""
    W, H, spp = 160, 90, 2
    try:
        scene, ntris = S.instanced_scene(pkg, W, H, 116)
        assert ntris == 116 * 868352
        org, dir_ = S.ray_batch(scene, W, H, seed=12)
        ctx.upload(scene)
        t, prim, _ = ctx.intersect(org, dir_)
        ot, oprim, _, _ = O.intersect(S.as_oracle_desc(scene), org, dir_)
        check_hits(t, prim, ot, oprim, "100.7 M triangles")
        assert np.array_equal(t, ot)
        hit_mesh = (oprim != O.PRIM_MISS) & (oprim < ntris)
        assert hit_mesh.mean() > 0.2 and oprim[hit_mesh].max() > ntris // 2      # copies far down the prim id range are hit
        ctx.film_bind(W, H)
        ctx.render(0, spp, spp, seed=12)
        film = ctx.film_download()
        ofilm, _ = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=12)
        assert np.isfinite(film).all() and (film >= 0).all()
        assert_same_film(film, ofilm, "100.7 M triangles")
    finally:
        S._INSTANCED.clear()                      # 10 GB of host buffers
