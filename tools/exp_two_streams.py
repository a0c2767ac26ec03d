"""Experiment: one frame rendered by K contexts (K streams) on one GPU at once, each taking 1/K of the samples."""
import os, sys, time, threading
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as G, scenes as S
pkg = G.load_package()
W, H, SPP = 1920, 1080, 512
scene = pkg.HostScene(S.scene_file(pkg, sys.argv[2] if len(sys.argv) > 2 else "dragon"), W, H)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ctxs = []
for k in range(K):
    c = pkg.Context(0); c.upload(scene); c.film_bind(W, H); ctxs.append(c)
def run(seed):
    stats = [None] * K
    def work(k):
        b, n = pkg.spp_partition(SPP, K, k)
        stats[k] = ctxs[k].render(b, n, SPP, seed=seed)
    th = [threading.Thread(target=work, args=(k,)) for k in range(K)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    return (time.perf_counter() - t0) * 1e3, sum(s.rays for s in stats)
run(0)
for i in range(3):
    ms, rays = run(1 + i)
    print("K=%d  %.2f ms  %.1f Mrays/s" % (K, ms, rays / ms / 1e3))
