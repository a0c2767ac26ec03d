"""Worker for the two-rank gloo test: renders this rank's share of 6 spp with the CPU oracle standing in for the device
renderer, reduces the films through the package's distributed helper, and rank 0 saves the result."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as G  # noqa: E402
import oracle_lib as O  # noqa: E402
import scenes as S  # noqa: E402


def main():
    pkg = G.load_package()
    import cuda_path_tracer_amd.distributed as D

    rank, world, _ = D.init_from_env(backend="gloo")
    assert world == 2
    W, H, spp = 48, 32, 6
    scene = S.spheres_scene(pkg, W, H)
    film = torch.zeros((H, W, 3), dtype=torch.float32)

    def render_fn(begin, count, total, seed):
        arr = film.numpy()
        O.render(S.as_oracle_desc(scene), W, H, begin, count, total, seed=seed, nthreads=2, film=arr)
        return count

    count = D.render_and_reduce(render_fn, film, spp, seed=4)
    assert count == 3
    if rank == 0:
        np.save(sys.argv[1], film.numpy())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
