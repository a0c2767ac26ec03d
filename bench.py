"""Benchmark: Mrays/s and ms/frame of the wavefront render hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one full frame of the headline workload: the reference's scene/dragon.scene camera, plane and glossy mesh
(the missing dragon.obj / forest.hdr replaced by the deterministic stand-ins: 868,352 triangles, 2048x1024 map) at
1920x1080, 512 samples per pixel.  With N > 1 every rank holds the whole scene, renders spp/N samples of every pixel
and the float films are summed onto rank 0 with one reduce (RCCL); the reduce is inside the timed region.  Total work
is fixed, so scaling is "strong".  One ray = one closest-hit query.

The JSON line also carries
  roofline      the extend (traversal) kernel against the 8 TB/s HBM peak: algorithmic bytes per ray (120 B of path
                state + 64 B per inner node visited + 48 B per triangle tested + 16 B per leaf entered, the per-ray
                means counted by the CPU oracle on a reduced-resolution run of the same scene) times the rays of the
                timed steps, divided by the HIP-event time of the extend launches on the library's own stream;
  cpu_baseline  the CPU oracle (a host restatement of the reference's traversal and shading) timed on this machine's
                cores on a bounded sample of the same workload (960x540, 320 spp).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import __graft_entry__ as G  # noqa: E402

HBM_PEAK_GBS = 8000.0
B_STATE = 120.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=512)
    ap.add_argument("--scene", default="dragon", choices=["dragon", "bunny", "balls", "city", "synth"])
    ap.add_argument("--instances", type=int, default=116,
                    help="--scene synth: copies of the dragon stand-in flattened into one mesh (116 = 100.7 M triangles)")
    ap.add_argument("--pool", type=int, default=0, help="path slots in flight (0: library default)")
    ap.add_argument("--extend-mode", type=int, default=2, help="2 persistent single-step waves (default), 1 loop-per-phase batches")
    ap.add_argument("--tune", default="", help="comma list of option=value pairs passed to pt_set_option")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    pkg = G.load_package()
    import cuda_path_tracer_amd.distributed as D
    import scenes as S

    # PT_BENCH_REHEARSAL=1: every rank on GPU 0 and gloo for the collectives, to run the N > 1 code path on a one-GPU box
    rehearsal = os.environ.get("PT_BENCH_REHEARSAL") == "1"
    rank, world, local_rank = D.init_from_env("gloo" if rehearsal else None)
    if rehearsal:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    # scene assembly (host, untimed): rank 0 generates the cached assets, the others wait
    file_scene = "dragon" if args.scene == "synth" else args.scene
    if rank == 0:
        scene_path = S.scene_file(pkg, file_scene)
    if world > 1:
        dist.barrier()
    scene_path = S.scene_file(pkg, file_scene)

    def make_scene(w, h):
        if args.scene == "synth":
            return S.instanced_scene(pkg, w, h, args.instances)[0]
        return pkg.HostScene(scene_path, w, h)

    t0 = time.time()
    scene = make_scene(args.width, args.height)
    t_scene = time.time() - t0

    ctx = pkg.Context(local_rank)
    if args.pool:
        ctx.set_option(pkg.OPT_POOL_SIZE, args.pool)
    ctx.set_option(pkg.OPT_EXTEND_MODE, args.extend_mode)
    streams = 3                                    # the library's default
    for kv in filter(None, args.tune.split(",")):
        k, v = kv.split("=")
        ctx.set_option(int(k), int(v))
        if int(k) == pkg.OPT_STREAMS:
            streams = int(v)
    t0 = time.time()
    ctx.upload(scene)
    t_upload = time.time() - t0
    if rank == 0:
        print("scene assembly %.2f s, upload %.2f s (host side, outside the timed region)" % (t_scene, t_upload),
              file=sys.stderr)
    film = torch.zeros((args.height, args.width, 3), dtype=torch.float32, device=dev)
    ctx.film_bind(args.width, args.height, film.data_ptr())
    torch.cuda.synchronize()

    acc = {"rays": 0, "rays_escaped": 0, "ms_extend": 0.0, "ms_extend_busy": 0.0, "launches": 0, "ms_render": 0.0}

    def render_fn(begin, count, total, seed):
        st = ctx.render(begin, count, total, seed)
        acc["rays"] += st.rays
        acc["rays_escaped"] += st.rays_escaped
        acc["ms_extend"] += st.ms_extend
        acc["ms_extend_busy"] += st.ms_extend_busy
        acc["launches"] += st.iterations
        acc["ms_render"] += st.ms_total
        return st

    def step(i):
        film.zero_()
        torch.cuda.synchronize()
        D.render_and_reduce(render_fn, film, args.spp, seed=i)

    for i in range(args.warmup):
        step(i)
    acc = {k: 0 for k in acc}
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(1000 + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    # whole-job numbers: max time over ranks, rays summed over ranks
    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    rr = torch.tensor([float(acc["rays"])], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
    elapsed = float(tt.item())
    total_rays = float(rr.item())

    if rank == 0:
        ms_per_step = elapsed * 1e3 / max(1, args.steps)
        mrays = total_rays / elapsed / 1e6
        out = {
            "metric": "Mrays/s (ms/frame in ms_per_step), %s %dx%d %dspp" % (args.scene, args.width, args.height, args.spp),
            "value": round(mrays, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s %dx%d %d spp (procedural stand-ins for the unshipped dragon.obj and .hdr maps)"
                            % ("%d-instance dragon stand-in" % args.instances if args.scene == "synth" else args.scene + ".scene",
                               args.width, args.height, args.spp),
                "triangles": int(scene.desc.num_tris),
                "rays_per_frame": int(total_rays / max(1, args.steps)),
                "parallelism": "spp-partition x%d + film reduce" % world,
                "scene_build_s": round(t_scene, 2),
            },
        }
        # CPU oracle on a bounded sample: gives the baseline and the per-ray traversal means for the roofline
        cpu = None
        b_ray = None
        if not args.no_cpu_baseline and world == 1:
            import oracle_lib as O

            cw, ch, cspp = 960, 540, 320      # about 10 s on 16 cores
            cscene = make_scene(cw, ch)
            threads = min(os.cpu_count() or 1, 16)
            _, ost = O.render(S.as_oracle_desc(cscene), cw, ch, 0, cspp, cspp, seed=1, nthreads=threads)
            cpu = {
                "value": round(ost.rays / ost.ms / 1e3, 3),
                "unit": "Mrays/s",
                "cores": threads,
                "kind": "port",
                "sample": "%s.scene %dx%d %d spp: %d rays in %.0f ms" % (args.scene, cw, ch, cspp, ost.rays, ost.ms),
            }
            b_ray = B_STATE + (64.0 * ost.nodes + 48.0 * ost.tris + 16.0 * ost.leaves) / max(1, ost.rays)
            out["config"]["traversal_means"] = {"nodes": round(ost.nodes / ost.rays, 3),
                                                "tris": round(ost.tris / ost.rays, 3),
                                                "leaves": round(ost.leaves / ost.rays, 3),
                                                "bytes_per_ray": round(b_ray, 1)}
        out["cpu_baseline"] = cpu
        # HBM-side bytes per extend launch, from committed rocprofv3 --pmc passes of this same command (not collected here:
        # counter collection needs its own profiler runs); null when the summary is absent or is for another workload
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r1_extend_traffic.json")
        if os.path.exists(tpath) and args.scene == "dragon" and (args.width, args.height, args.spp) == (1920, 1080, 512):
            prof = json.load(open(tpath))
            if world == 1:
                traffic = prof.get("traffic_bytes_per_launch")
            if b_ray is None:
                # the oracle's per-ray traversal means of this workload are deterministic; without the CPU leg (N > 1 or
                # --no-cpu-baseline) the committed figure of the same workload is used
                b_ray = prof.get("algorithmic_bytes_per_ray")
        if b_ray is not None and acc["ms_extend"] > 0:
            # b_ray is the mean over ALL rays.  Rays that escaped where they were created (one root visit in the oracle's
            # count) never reach an extend launch, so their bytes are taken out of what the extend launches are credited with.
            has_tree = int(scene.desc.num_nodes) > 0
            ext_bytes = b_ray * acc["rays"] - acc["rays_escaped"] * (B_STATE + (64.0 if has_tree else 0.0))
            ext_rays = acc["rays"] - acc["rays_escaped"]
            achieved = ext_bytes / (acc["ms_extend"] * 1e-3) / 1e9
            out["config"]["rays_escaped_at_creation_per_frame"] = int(acc["rays_escaped"] / max(1, args.steps))
            out["roofline"] = {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel": "k_extend",
                "avg_launch_ms": round(acc["ms_extend"] / max(1, acc["launches"]), 4),
                "launches": int(acc["launches"]),
                "rays_per_launch": round(ext_rays / max(1, acc["launches"]), 1),
                "bytes_per_launch": round(ext_bytes / max(1, acc["launches"]), 0),
                # The library runs several wavefront loops on streams of their own, so extend launches overlap each other
                # (and the other kernels): a launch then has only part of the device and its own duration understates what
                # the device does.  achieved_all_streams divides the same bytes by the time during which at least one
                # extend launch was running.
                "concurrent_streams": streams,
                "achieved_all_streams": round(ext_bytes / (acc["ms_extend_busy"] * 1e-3) / 1e9, 1),
                "frac_all_streams": round(ext_bytes / (acc["ms_extend_busy"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "extend_busy_share_of_frame": round(acc["ms_extend_busy"] / max(1e-9, acc["ms_render"]), 3),
            }
        else:
            out["roofline"] = None
        print(json.dumps(out), flush=True)

    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
