"""Freezes one small film of the CPU oracle into tests/golden/film_knot.npz.

The film pins oracle + scene construction + the counter-based random streams against drift between rounds; the GPU test
compares the HIP path's film with the same file under the stated film tolerance.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as G  # noqa: E402
import oracle_lib as O  # noqa: E402
import scenes as S  # noqa: E402

W, H, SPP, SEED = 64, 40, 8, 17


def main():
    pkg = G.load_package()
    scene = S.knot_scene(pkg, W, H)
    film, st = O.render(S.as_oracle_desc(scene), W, H, 0, SPP, SPP, seed=SEED)
    path = os.path.join(ROOT, "tests", "golden", "film_knot.npz")
    np.savez_compressed(path, film=film, width=W, height=H, spp=SPP, seed=SEED, rays=np.uint64(st.rays),
                        paths=np.uint64(st.paths))
    print(path, film.shape, "rays", st.rays, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
