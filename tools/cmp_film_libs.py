import sys, importlib, os
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import scenes as S
films = []
for lib in sys.argv[1:]:
    os.environ["PT_HIP_LIB"] = os.path.abspath(lib)
    for m in [m for m in sys.modules if m.startswith("cuda-path-tracer_amd")]:
        del sys.modules[m]
    pkg = importlib.import_module("cuda-path-tracer_amd")
    pkg._hip = None
    ctx = pkg.Context(0)
    sc = S.knot_scene(pkg, 160, 96)
    ctx.upload(sc); ctx.film_bind(160, 96); ctx.render(0, 16, 16, seed=3)
    films.append(ctx.film_download()); ctx.close()
print("films bit-identical:", np.array_equal(films[0], films[1]), "max abs diff", float(np.abs(films[0]-films[1]).max()))
