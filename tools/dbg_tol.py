import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as G, oracle_lib as O, scenes as S
pkg = G.load_package()
ctx = pkg.Context(0)
for name, W, H, spp in [("knot", 96, 64, 8), ("balls", 96, 64, 8), ("knot", 64, 48, 1024), ("knot", 1920, 1080, 2)]:
    scene = S.knot_scene(pkg, W, H) if name == "knot" else S.spheres_scene(pkg, W, H)
    ctx.upload(scene); ctx.film_bind(W, H)
    ctx.render(0, spp, spp, seed=5); film = ctx.film_download()
    ofilm, _ = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=5, nthreads=16)
    ofilm2, _ = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=6, nthreads=16)
    d = film.astype(np.float64) - ofilm
    differing = (np.abs(d).max(axis=2) > 1e-3 * (1e-3 + np.abs(ofilm).max(axis=2))).mean()
    relmse = np.mean(d * d / (ofilm.astype(np.float64) ** 2 + 1e-2))
    d2 = ofilm2.astype(np.float64) - ofilm
    floor = np.mean(d2 * d2 / (ofilm.astype(np.float64) ** 2 + 1e-2))
    print(name, W, H, spp, "differing %.4f relMSE %.3g floor %.3g ratio %.4f" % (differing, relmse, floor, relmse / floor))
