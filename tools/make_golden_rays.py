"""Freezes closest-hit answers of the CPU oracle on fixed ray batches into tests/golden/rays_*.npz.

Each file holds the ray batch (origins, directions) and the oracle's hit distance and primitive id for one scene.
The scenes are rebuilt from tests/scenes.py, so the fixture pins oracle + scene construction against drift; the GPU
tests compare the HIP path with the same files.
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


def main():
    pkg = G.load_package()
    W, H = 48, 32
    cases = S.parity_cases(pkg, W, H)
    for name, sc in cases.items():
        org, dir_ = S.ray_batch(sc, W, H, seed=11)
        t, prim, _, _ = O.intersect(S.as_oracle_desc(sc), org, dir_)
        path = os.path.join(ROOT, "tests", "golden", "rays_%s.npz" % name)
        np.savez_compressed(path, org=org, dir=dir_, t=t, prim=prim)
        hits = (prim != 0xFFFFFFFF).mean()
        print(name, org.shape[0], "rays, hit rate %.2f" % hits, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
