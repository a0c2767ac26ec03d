import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as G, oracle_lib as O, scenes as S
pkg = G.load_package()
W, H = 96, 64
name = sys.argv[1] if len(sys.argv) > 1 else "knot"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
scene = S.knot_scene(pkg, W, H) if name == "knot" else S.spheres_scene(pkg, W, H)
ctx = pkg.Context(0); ctx.upload(scene); ctx.film_bind(W, H)
st = ctx.render(0, spp, spp, seed=5)
film = ctx.film_download()
ofilm, ost = O.render(S.as_oracle_desc(scene), W, H, 0, spp, spp, seed=5)
print("rays gpu", st.rays, "oracle", ost.rays, "iters", st.iterations)
d = np.abs(film.astype(np.float64) - ofilm)
rel = d.max(axis=2) / (1e-6 + np.abs(ofilm).max(axis=2))
print("pixels differing rel>1e-4: %.4f  >1e-2: %.4f" % ((rel > 1e-4).mean(), (rel > 1e-2).mean()))
print("relMSE", np.mean(d * d / (ofilm.astype(np.float64) ** 2 + 1e-2)))
idx = np.argsort(rel.ravel())[::-1][:12]
for k in idx:
    y, x = divmod(int(k), W)
    print((x, y), "gpu", film[y, x], "oracle", ofilm[y, x], "rel %.3g" % rel[y, x])
print("nan gpu", np.isnan(film).sum(), "nan oracle", np.isnan(ofilm).sum())
