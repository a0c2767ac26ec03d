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
