"""Closest-hit query rates of the two boundaries on one workload: pt_intersect (host buffers: the rays cross PCIe in, the
answers cross it back) against pt_intersect_device (everything resident).  Rays: the camera rays of a 3840x2160 frame of the
headline scene.  usage: python tools/intersect_rates.py"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import scenes as S  # noqa: E402

pkg = importlib.import_module("cuda-path-tracer_amd")


def main():
    W, H = 3840, 2160
    scene = pkg.HostScene(S.scene_file(pkg, "dragon"), W, H)
    ctx = pkg.Context(0)
    ctx.upload(scene)
    org, dir_ = ctx.primary_rays(W, H, sample=0, seed=1)
    n = org.shape[0]
    reps = 5
    ctx.intersect(org[:1000], dir_[:1000])
    t0 = time.perf_counter()
    for _ in range(reps):
        t, prim, _ = ctx.intersect(org, dir_)
    host_s = (time.perf_counter() - t0) / reps
    o4 = torch.zeros((n, 4), dtype=torch.float32)
    d4 = torch.zeros((n, 4), dtype=torch.float32)
    o4[:, :3] = torch.from_numpy(org)
    d4[:, :3] = torch.from_numpy(dir_)
    o4, d4 = o4.cuda(), d4.cuda()
    hit = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    td = torch.zeros((n,), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctx.intersect_device(n, o4.data_ptr(), d4.data_ptr(), hit.data_ptr(), td.data_ptr())
    t0 = time.perf_counter()
    for _ in range(reps):
        st = ctx.intersect_device(n, o4.data_ptr(), d4.data_ptr(), hit.data_ptr(), td.data_ptr())
    torch.cuda.synchronize()
    dev_s = (time.perf_counter() - t0) / reps
    same = bool(np.array_equal(td.cpu().numpy(), t) and np.array_equal(hit[:, 3].cpu().numpy().view(np.uint32), prim))
    moved = n * (12 + 12 + 4 + 4)
    print("%d camera rays of the %dx%d headline frame, answers identical: %s" % (n, W, H, same))
    print("pt_intersect (host buffers, %.0f MB over PCIe per call): %.2f ms = %.1f Mrays/s" % (moved / 1e6, host_s * 1e3, n / host_s / 1e6))
    print("pt_intersect_device (resident):                           %.2f ms = %.1f Mrays/s (kernel %.2f ms)" %
          (dev_s * 1e3, n / dev_s / 1e6, st.ms_extend))
    ctx.close()


if __name__ == "__main__":
    main()
