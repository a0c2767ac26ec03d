"""The closest-hit entry point that allocates nothing (pt_intersect_device_noalloc) against pt_intersect_device.

Both run the same kernels; the first keeps its hit-id scratch and counters in a static device array.  Two calls in a row,
the second with fewer rays, so that what the first call left in the array would show if the second read any of it.
"""
import os

import numpy as np
import pytest

import scenes as S

pytestmark = pytest.mark.gpu


def _device_rays(org, dir_):
    import torch

    n = org.shape[0]
    o4 = torch.zeros((n, 4), dtype=torch.float32)
    d4 = torch.zeros((n, 4), dtype=torch.float32)
    o4[:, :3] = torch.from_numpy(org)
    d4[:, :3] = torch.from_numpy(dir_)
    return o4.cuda(), d4.cuda()


def _run(fn, o4, d4):
    import torch

    n = o4.shape[0]
    hit = torch.full((n, 4), -7.0, dtype=torch.float32, device="cuda")
    t = torch.full((n,), -7.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    st = fn(n, o4.data_ptr(), d4.data_ptr(), hit.data_ptr(), t.data_ptr())
    return hit.cpu().numpy(), t.cpu().numpy(), st


def test_no_alloc_entry_point_matches_allocating_one(pkg, ctx):
    g = np.load(os.path.join(S.GOLDEN, "rays_knot.npz"))
    scene = S.parity_cases(pkg)["knot"]
    ctx.upload(scene)
    ctx.set_option(pkg.OPT_COUNT_TRAVERSAL, 1)
    try:
        big_o, big_d = np.tile(g["org"], (50, 1)), np.tile(g["dir"], (50, 1))     # persistent waves refill many times
        small_o, small_d = g["org"][:777][::-1].copy(), g["dir"][:777][::-1].copy()
        for org, dir_ in ((big_o, big_d), (small_o, small_d)):
            o4, d4 = _device_rays(org, dir_)
            hit_a, t_a, st_a = _run(ctx.intersect_device, o4, d4)
            hit_b, t_b, st_b = _run(ctx.intersect_device_noalloc, o4, d4)
            assert np.array_equal(t_a, t_b)
            assert np.array_equal(hit_a.view(np.uint32), hit_b.view(np.uint32))      # normals and primitive ids, bit for bit
            assert (st_a.nodes, st_a.tris, st_a.leaves) == (st_b.nodes, st_b.tris, st_b.leaves)
            assert st_b.rays == org.shape[0]
        # and the fixture itself: the first call's answers are the frozen ones
        o4, d4 = _device_rays(g["org"], g["dir"])
        hit, t, _ = _run(ctx.intersect_device_noalloc, o4, d4)
        assert np.array_equal(t, g["t"])
        assert np.array_equal(hit[:, 3].view(np.uint32), g["prim"])
    finally:
        ctx.set_option(pkg.OPT_COUNT_TRAVERSAL, 0)


def test_no_alloc_entry_point_rejects_what_its_scratch_cannot_hold(pkg, ctx):
    import torch

    ctx.upload(S.spheres_scene(pkg, 48, 32))
    n = (1 << 22) + 1
    buf = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    t = torch.zeros((n,), dtype=torch.float32, device="cuda")
    with pytest.raises(pkg.PtError, match="static scratch"):
        ctx.intersect_device_noalloc(n, buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), t.data_ptr())
    st = ctx.intersect_device_noalloc(n - 1, buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), t.data_ptr())
    assert st.rays == n - 1


def test_no_alloc_entry_point_allocates_no_device_memory(pkg):
    """The property the entry point exists for, on a fresh context whose first closest-hit call it is: the device's free
    memory is the same before and after (hipMemGetInfo); the allocating entry point, called next with the same rays, does
    take memory (its hit-id buffer), which shows that the measurement would see an allocation."""
    import torch

    c = pkg.Context(0)
    try:
        c.upload(S.parity_cases(pkg)["knot"])
        g = np.load(os.path.join(S.GOLDEN, "rays_knot.npz"))
        reps = (1 << 22) // g["org"].shape[0]                    # close to the 4 Mi rays the scratch holds: 16 MiB of hit ids
        o4, d4 = _device_rays(np.tile(g["org"], (reps, 1)), np.tile(g["dir"], (reps, 1)))
        n = o4.shape[0]
        hit = torch.empty((n, 4), dtype=torch.float32, device="cuda")
        t = torch.empty((n,), dtype=torch.float32, device="cuda")
        # 64 rays first: whatever the HIP runtime itself sets up at a kernel's first launch is not this call's doing (a
        # buffer of the library's own would grow with the ray count and show in the 4 Mi-ray calls below)
        c.intersect_device_noalloc(64, o4.data_ptr(), d4.data_ptr(), hit.data_ptr(), t.data_ptr())
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        c.intersect_device_noalloc(n, o4.data_ptr(), d4.data_ptr(), hit.data_ptr(), t.data_ptr())
        c.intersect_device_noalloc(n, o4.data_ptr(), d4.data_ptr(), hit.data_ptr(), t.data_ptr(), want_stats=False)
        torch.cuda.synchronize()
        free1 = torch.cuda.mem_get_info()[0]
        assert free1 == free0, "pt_intersect_device_noalloc changed the device's free memory by %d bytes" % (free0 - free1)
        c.intersect_device(n, o4.data_ptr(), d4.data_ptr(), hit.data_ptr(), t.data_ptr())
        assert torch.cuda.mem_get_info()[0] < free0
    finally:
        c.close()


def test_no_alloc_entry_point_queued_form(pkg, ctx):
    """stats == NULL: the call only queues its two kernels on the context's stream (no memset, event, copy or wait); after
    the stream has been waited for, the results are those of the synchronous form."""
    import torch

    g = np.load(os.path.join(S.GOLDEN, "rays_knot.npz"))
    ctx.upload(S.parity_cases(pkg)["knot"])
    o4, d4 = _device_rays(np.tile(g["org"], (20, 1)), np.tile(g["dir"], (20, 1)))
    n = o4.shape[0]
    hit = torch.full((n, 4), -7.0, dtype=torch.float32, device="cuda")
    t = torch.full((n,), -7.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    assert ctx.intersect_device_noalloc(n, o4.data_ptr(), d4.data_ptr(), hit.data_ptr(), t.data_ptr(), want_stats=False) is None
    torch.cuda.synchronize()                                      # device-wide: includes the context's own stream
    assert np.array_equal(t.cpu().numpy(), np.tile(g["t"], 20))
    assert np.array_equal(hit.cpu().numpy()[:, 3].view(np.uint32), np.tile(g["prim"], 20))


def test_no_alloc_entry_point_on_deep_trees(pkg, ctx):
    """A chain-shaped BVH 40 levels deep: deeper than the 26 stack slots the allocating entry point keeps in LDS (it sends the
    rest through its global overflow array), while the variant, which has no such array, holds all 41 slots in LDS.  Rays
    along the chain fill the stack to the bottom.  Both equal the oracle.  A tree of 50 levels needs 51 slots, more than
    the 48 the variant's LDS-only stack may take: refused with PT_ERR_ARG, and still traced by the allocating entry point."""
    import oracle_lib as O

    for levels, fits in ((40, True), (50, False)):
        scene = S.chain_scene(pkg, levels)
        org, dir_ = S.chain_rays(levels, n=8192, seed=levels)
        ot, oprim, _, _ = O.intersect(S.as_oracle_desc(scene), org, dir_)
        assert (oprim == levels).mean() > 0.5                    # the rays that walk the whole chain hit its last triangle
        ctx.upload(scene)
        o4, d4 = _device_rays(org, dir_)
        hit_a, t_a, _ = _run(ctx.intersect_device, o4, d4)
        assert np.array_equal(t_a, ot) and np.array_equal(hit_a[:, 3].view(np.uint32), oprim)
        if fits:
            hit_b, t_b, _ = _run(ctx.intersect_device_noalloc, o4, d4)
            assert np.array_equal(t_b, ot)
            assert np.array_equal(hit_a.view(np.uint32), hit_b.view(np.uint32))
        else:
            with pytest.raises(pkg.PtError, match="stack slots"):
                _run(ctx.intersect_device_noalloc, o4, d4)


def test_no_alloc_entry_point_under_stream_capture(pkg, ctx):
    """The queued form (stats == NULL) inside a HIP stream capture of the context's stream: the capture takes the call (it
    allocates nothing, copies nothing back and waits for nothing), and launching the captured graph gives the answers of the
    synchronous form."""
    import ctypes as C

    import torch

    hip = C.CDLL("libamdhip64.so")          # the runtime already in the process (torch loaded it)
    for f in ("hipStreamBeginCapture", "hipStreamEndCapture", "hipGraphInstantiate", "hipGraphLaunch", "hipStreamSynchronize",
              "hipGraphExecDestroy", "hipGraphDestroy"):
        getattr(hip, f).restype = C.c_int
    g = np.load(os.path.join(S.GOLDEN, "rays_knot.npz"))
    ctx.upload(S.parity_cases(pkg)["knot"])
    o4, d4 = _device_rays(np.tile(g["org"], (8, 1)), np.tile(g["dir"], (8, 1)))
    n = o4.shape[0]
    hit = torch.full((n, 4), -7.0, dtype=torch.float32, device="cuda")
    t = torch.full((n,), -7.0, dtype=torch.float32, device="cuda")
    # one ordinary call first: a kernel's first launch loads its code object, which is not a thing to do under capture
    ctx.intersect_device_noalloc(n, o4.data_ptr(), d4.data_ptr(), hit.data_ptr(), t.data_ptr())
    hit.fill_(-7.0)
    t.fill_(-7.0)
    torch.cuda.synchronize()
    stream = C.c_void_p(ctx.stream)
    graph, gexec = C.c_void_p(), C.c_void_p()
    assert hip.hipStreamBeginCapture(stream, 2) == 0                       # hipStreamCaptureModeRelaxed
    try:
        ctx.intersect_device_noalloc(n, o4.data_ptr(), d4.data_ptr(), hit.data_ptr(), t.data_ptr(), want_stats=False)
    finally:
        rc = hip.hipStreamEndCapture(stream, C.byref(graph))
    assert rc == 0 and graph.value, "the capture was refused or invalidated (%d)" % rc
    torch.cuda.synchronize()
    assert (t.cpu().numpy() == -7.0).all(), "the call ran instead of being captured"
    assert hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, C.c_size_t(0)) == 0
    try:
        for _ in range(2):                                                  # a graph is there to be launched repeatedly
            hit.fill_(-7.0)
            t.fill_(-7.0)
            torch.cuda.synchronize()
            assert hip.hipGraphLaunch(gexec, stream) == 0
            assert hip.hipStreamSynchronize(stream) == 0
            assert np.array_equal(t.cpu().numpy(), np.tile(g["t"], 8))
            assert np.array_equal(hit.cpu().numpy()[:, 3].view(np.uint32), np.tile(g["prim"], 8))
    finally:
        hip.hipGraphExecDestroy(gexec)
        hip.hipGraphDestroy(graph)
    # and the context is usable afterwards
    st = ctx.intersect_device_noalloc(n, o4.data_ptr(), d4.data_ptr(), hit.data_ptr(), t.data_ptr())
    assert st.rays == n and np.array_equal(t.cpu().numpy(), np.tile(g["t"], 8))
