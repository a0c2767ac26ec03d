"""Benchmark: Mrays/s and ms/frame of the wavefront render hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one full frame of the headline workload: the reference's scene/dragon.scene camera, plane and glossy mesh
(the missing dragon.obj / forest.hdr replaced by the deterministic stand-ins: 868,352 triangles, 2048x1024 map) at
1920x1080, 512 samples per pixel.  With N > 1 every rank holds the whole scene, renders spp/N samples of every pixel
and the float films are summed onto rank 0 with one reduce (RCCL); the reduce is inside the timed region.  Total work
is fixed, so scaling is "strong".  One ray = one closest-hit query.

`--gpus N` without a launcher (no WORLD_SIZE in the environment) starts the N ranks itself: N fresh child processes, one
per GPU, before this process has made any GPU call; rank 0's JSON line is relayed and the exit code is non-zero if any
rank failed.

The JSON line also carries
  roofline      the extend (traversal) kernel against the 8 TB/s HBM peak.  Headline (achieved, frac, traffic): the bytes
                the kernel really moves on the fabric side of the L2s, from committed rocprofv3 --pmc passes of this very
                command, over the kernel's run time in the same profile (launches serialised there) -- a figure that cannot
                exceed the peak.  It is reported only when the profile was taken on the kernel sources being run (a hash
                of csrc/ ties them); otherwise null.  Beside it, measured live in this run with HIP events on the library's
                own streams: `algorithmic` = the algorithmic bytes per launch (120 B of path state + 64 B per inner node
                visited + 48 B per triangle tested + 16 B per leaf entered, per-ray means counted by the CPU oracle on a
                reduced-resolution run of the same scene, times the rays the launch traces) over the mean duration of the
                non-empty extend launches.  The tree is served by L1 / L2 / Infinity Cache and launches of several streams
                overlap, so that figure is NOT an HBM fraction and may exceed the peak; `record_rate` = the 64 B records
                the launches fetch per second of extend time, next to the rate a bare dependent-gather probe reaches at the
                same cache-hit mix and number of fetching lanes (the unit that really binds the kernel is the CU's handling
                of per-lane fetches that miss L1; the probe is a reference for it, not a bound).
  cpu_baseline  the CPU oracle (a host restatement of the reference's traversal and shading) timed on this machine's
                cores on a bounded sample of the same workload (960x540, 320 spp).
  film_check    the film of the last timed frame: finite everywhere, and its mean next to the mean of the oracle's film
                of the bounded sample (same camera and scene at a quarter of the pixels); the run fails when they differ
                by more than 2 %.  The bounded sample itself is also rendered on the GPU (untimed) and must equal the
                oracle's film bit for bit, ray count included.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0
B_STATE = 120.0
# committed counter-side profiles of the extend kernel, one per workload they were taken on
PROFILES = {"dragon": "profiles/r3_extend_traffic.json", "dragon_dense": "profiles/r3_extend_traffic_dragon_dense.json",
            "synth": "profiles/r3_extend_traffic_synth.json"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=512)
    ap.add_argument("--scene", default="dragon", choices=["dragon", "dragon_dense", "bunny", "balls", "city", "synth"],
                    help="dragon_dense: the dragon scene with the frame-filling variant of the stand-in mesh (91 %% of the "
                         "camera rays hit it, 44 inner nodes per ray)")
    ap.add_argument("--instances", type=int, default=116,
                    help="--scene synth: copies of the dragon stand-in flattened into one mesh (116 = 100.7 M triangles)")
    ap.add_argument("--pool", type=int, default=0, help="path slots in flight (0: library default)")
    ap.add_argument("--extend-mode", type=int, default=2, help="2 persistent single-step waves (default), 1 loop-per-phase batches")
    ap.add_argument("--tune", default="", help="comma list of option=value pairs passed to pt_set_option")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def launch_ranks(n):
    """Runs this command as n ranks, one fresh child process each, and returns the exit code for the parent.  Called before
    the parent has imported torch or touched the GPU: the children are ordinary child processes (nothing is re-executed),
    each gets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torch.distributed.run would set them.  Rank 0's
    standard output (the JSON line) is relayed; the other ranks' goes to standard error."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    try:
        return wait_for_ranks(procs)
    except BaseException:
        # interrupted: the ranks (this process's own children, by their process ids) do not outlive it
        for p in procs:
            if p.poll() is None:
                p.kill()
        raise


def wait_for_ranks(procs):
    n = len(procs)
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    # Wait for all ranks.  Once one has failed the others cannot finish their collectives: they get 30 s, then they are
    # ended (by their process ids, which this process owns).
    codes = [None] * n
    failed_at = None
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None and p.poll() is not None:
                codes[r] = p.returncode
                if codes[r] != 0 and failed_at is None:
                    failed_at = time.time()
        if failed_at is not None and time.time() - failed_at > 30:
            for r, p in enumerate(procs):
                if codes[r] is None:
                    p.kill()
                    codes[r] = p.wait()
        time.sleep(0.05)
    reader.join(timeout=10)
    failed = any(c != 0 for c in codes)
    out0 = out0[0] if out0 else b""
    # the JSON line to standard output, anything else a rank 0 library printed there (gloo does) to standard error
    for line in out0.decode(errors="replace").splitlines():
        print(line, file=sys.stdout if line.startswith("{") else sys.stderr)
    sys.stdout.flush()
    if failed:
        print("bench: rank exit codes %r" % (codes,), file=sys.stderr)
        return 1
    return 0


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    import __graft_entry__ as G

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
    modes = set()

    def render_fn(begin, count, total, seed):
        st = ctx.render(begin, count, total, seed)
        acc["rays"] += st.rays
        acc["rays_escaped"] += st.rays_escaped
        acc["ms_extend"] += st.ms_extend
        acc["ms_extend_busy"] += st.ms_extend_busy
        acc["launches"] += st.extend_launches
        acc["ms_render"] += st.ms_total
        modes.add(ctx.render_mode_used)
        return st

    def step(i):
        film.zero_()
        torch.cuda.synchronize()
        D.render_and_reduce(render_fn, film, args.spp, seed=i)

    if world > 1:
        # the first collective on a communicator sets it up (hundreds of ms): do that outside the timed region even when
        # no warm-up step was asked for
        dist.reduce(film, dst=0, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    acc = {k: 0 for k in acc}
    modes.clear()
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
    tt = torch.tensor([elapsed, acc["ms_extend_busy"], acc["ms_render"]], dtype=torch.float64, device=dev)
    keys = ["rays", "rays_escaped", "ms_extend", "launches"]
    rr = torch.tensor([float(acc[k]) for k in keys], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
    elapsed = float(tt[0].item())
    # rays, launches and the launches' summed durations over all ranks; busy and render time of the slowest rank
    acc.update({k: float(v) for k, v in zip(keys, rr.tolist())})
    acc["ms_extend_busy"], acc["ms_render"] = float(tt[1].item()), float(tt[2].item())
    total_rays = acc["rays"]
    film_host = film.cpu().numpy() if rank == 0 else None

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
                # 1: wavefront loop (bounce / generate / extend kernels over path buffers), 2: whole paths in one persistent
                # kernel; chosen by the library from the scene unless --tune 13= forces one
                "render_mode": sorted(modes)[0] if len(modes) == 1 else sorted(modes),
                "scene_build_s": round(t_scene, 2),
            },
        }
        # CPU oracle on a bounded sample: gives the baseline and the per-ray traversal means for the roofline
        cpu = None
        b_ray = None
        oracle_mean = None
        same_bits = None
        trav = None
        if not args.no_cpu_baseline and world == 1:
            import oracle_lib as O

            cw, ch, cspp = 960, 540, 320      # about 10 s on 16 cores
            cscene = make_scene(cw, ch)
            threads = min(os.cpu_count() or 1, 16)
            ofilm, ost = O.render(S.as_oracle_desc(cscene), cw, ch, 0, cspp, cspp, seed=1, nthreads=threads)
            oracle_mean = float(ofilm.astype(np.float64).mean())
            # the same frame on the GPU (same size, samples and seed; untimed): film and ray count must equal the
            # oracle's exactly
            gctx = pkg.Context(local_rank)
            gctx.upload(cscene)
            gctx.film_bind(cw, ch)
            gst = gctx.render(0, cspp, cspp, seed=1)
            gfilm = gctx.film_download()
            gctx.close()
            same_bits = bool(np.array_equal(gfilm.view(np.uint32), ofilm.view(np.uint32)) and gst.rays == ost.rays)
            cpu = {
                "value": round(ost.rays / ost.ms / 1e3, 3),
                "unit": "Mrays/s",
                "cores": threads,
                "kind": "port",
                "sample": "%s.scene %dx%d %d spp: %d rays in %.0f ms" % (args.scene, cw, ch, cspp, ost.rays, ost.ms),
            }
            b_ray = B_STATE + (64.0 * ost.nodes + 48.0 * ost.tris + 16.0 * ost.leaves) / max(1, ost.rays)
            trav = (ost.nodes / ost.rays, ost.tris / ost.rays)
            out["config"]["traversal_means"] = {"nodes": round(ost.nodes / ost.rays, 3),
                                                "tris": round(ost.tris / ost.rays, 3),
                                                "leaves": round(ost.leaves / ost.rays, 3),
                                                "bytes_per_ray": round(b_ray, 1)}
        out["cpu_baseline"] = cpu
        # The film of the last timed frame (on rank 0 after the reduce): it must be finite, and its mean must agree with the
        # oracle's film of the bounded sample.
        finite = bool(np.isfinite(film_host).all())
        film_mean = float(film_host.astype(np.float64).mean())
        out["film_check"] = {"finite": finite, "min": float(film_host.min()), "mean": round(film_mean, 6),
                             "oracle_mean": None if oracle_mean is None else round(oracle_mean, 6),
                             "rel_diff": None if oracle_mean is None else round(abs(film_mean - oracle_mean) / oracle_mean, 5),
                             # the cpu_baseline sample (same scene at 960x540, 320 spp) rendered on the GPU as well: every
                             # pixel's bits and the ray count equal to the oracle's
                             "sample_bit_equal_to_oracle": same_bits}
        # Counter-side figures of the extend kernel, from the committed rocprofv3 --pmc passes of this same command (not
        # collected here: counters need profiler runs of their own).  Used only when they were taken on this workload AND
        # on the kernel sources this run executes; otherwise the counter-side fields are null.
        prof = None
        prof_why = "no committed counter profile for this workload"
        workload = {"scene": args.scene, "width": args.width, "height": args.height, "spp": args.spp}
        if args.scene == "synth":
            workload["instances"] = args.instances
        out["config"]["workload_key"] = workload
        pfile = PROFILES.get(args.scene)
        if pfile and os.path.exists(os.path.join(ROOT, pfile)):
            cand = json.load(open(os.path.join(ROOT, pfile)))
            if cand.get("workload") != workload:
                prof_why = "%s was taken on %r" % (pfile, cand.get("workload"))
            else:
                if b_ray is None:
                    # the oracle's per-ray traversal means of this workload are deterministic; without the CPU leg (N > 1
                    # or --no-cpu-baseline) the committed figures of the same workload are used
                    b_ray = cand.get("algorithmic_bytes_per_ray")
                    trav = (cand.get("nodes_per_ray"), cand.get("tris_per_ray"))
                if cand.get("kernel_source_hash") != G.kernel_source_hash():
                    prof_why = "%s was taken on other kernel sources (%s, running %s)" % (
                        pfile, str(cand.get("kernel_source_hash"))[:12], G.kernel_source_hash()[:12])
                elif args.tune or args.pool or args.extend_mode != 2:
                    prof_why = "%s was taken with the default options" % pfile
                elif cand.get("render_mode", 1) not in modes:
                    prof_why = "%s was taken in render mode %r" % (pfile, cand.get("render_mode", 1))
                else:
                    prof = cand
                    prof["file"] = pfile
        if b_ray is not None and acc["ms_extend"] > 0 and acc["launches"] > 0:
            # b_ray is the mean over ALL rays.  Rays that were settled where they were created (one root visit in the
            # oracle's count) never reach an extend launch, so their bytes are taken out of what the launches are credited with.
            has_tree = int(scene.desc.num_nodes) > 0
            whole_paths = modes == {2}          # the one kernel then handles every ray, the settled ones included
            kernel = "k_path_stream" if whole_paths else "k_extend_stream"
            settled_out = 0.0 if whole_paths else acc["rays_escaped"]
            ext_bytes = b_ray * acc["rays"] - settled_out * (B_STATE + (64.0 if has_tree else 0.0))
            ext_rays = acc["rays"] - settled_out
            launches = acc["launches"]
            avg_ms = acc["ms_extend"] / launches
            alg_launch = ext_bytes / (acc["ms_extend"] * 1e-3) / 1e9
            alg_frame = ext_bytes / elapsed / 1e9
            counters = prof is not None and world == 1
            traffic = prof.get("traffic_bytes_per_launch") if counters else None
            achieved = prof.get("achieved_traffic_gbs") if counters else None
            out["config"]["rays_escaped_at_creation_per_frame"] = int(acc["rays_escaped"] / max(1, args.steps))
            roof = {
                # Headline: bytes the kernel moves on the fabric side of the L2s (FETCH_SIZE x 2 + WRITE_SIZE, Infinity-Cache
                # hits included: an upper estimate of the HBM bytes) per launch, over the launch's duration with the launches
                # serialised, both from the committed profile of this command on these kernel sources.  Cannot exceed 1.
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel": kernel,
                "what_binds": "the rate at which a CU turns around per-lane 64 B record fetches that miss L1 (see record_rate); "
                              "HBM itself binds only where the tree does not fit the 256 MB Infinity Cache (--scene synth)",
                "counters_source": prof["file"] if counters else None,
                "counters_missing_because": None if counters else (prof_why if world == 1 else "N > 1"),
                # Measured live in this run (HIP events on the library's streams).  NOT an HBM fraction: the algorithmic bytes
                # are mostly served by L1 / L2 / Infinity Cache, and with several streams a launch's duration includes the
                # time it shares the GPU with other launches.  May exceed the peak.
                "algorithmic": {
                    "bytes_per_ray": round(b_ray, 1),
                    "launches": int(launches),
                    "avg_launch_ms": round(avg_ms, 4),
                    "rays_per_launch": round(ext_rays / launches, 1),
                    "bytes_per_launch": round(ext_bytes / launches, 0),
                    "gbs_per_launch": round(alg_launch, 1),
                    "frac_per_launch": round(alg_launch / HBM_PEAK_GBS, 4),
                    "gbs_over_frame": round(alg_frame, 1),
                    "over_frame": round(alg_frame / HBM_PEAK_GBS, 4),
                    "traffic_over_algorithmic": None if not traffic else round(traffic * launches / ext_bytes, 4),
                },
                "concurrent_streams": 1 if whole_paths else streams,
                "extend_busy_share_of_frame": round(acc["ms_extend_busy"] / max(1e-9, acc["ms_render"]), 3),
                # rays that enter an extend launch (the ones a tree is traversed for) per second of frame time
                "traversed_mrays_per_s": round(ext_rays / elapsed / 1e6, 1),
            }
            if trav and trav[0] is not None:
                # 64 B records the launches fetch (every node visit and triangle test of the rays they trace) per second of
                # the time at least one extend launch runs, against a bare dependent-gather loop at extend's own cache-hit
                # mix and lane occupancy (tools/gather_probe2, figure from the committed profile)
                recs = (trav[0] + trav[1]) * acc["rays"] - settled_out * (1.0 if has_tree else 0.0)
                rate = recs / (acc["ms_extend_busy"] * 1e-3)
                ceiling = None if not (counters and prof.get("probe_mixed_g_records_per_s")) else prof["probe_mixed_g_records_per_s"] * 1e9
                roof["record_rate"] = {
                    "records_per_s": round(rate, 0),
                    # a reference rate, not a bound: the probe's fetches are uniformly random within its three sets
                    "probe_at_kernel_mix_records_per_s": ceiling,
                    "ratio_to_probe": None if not ceiling else round(rate / ceiling, 4),
                    "kernel_alone_records_per_s": prof.get("extend_alone_records_per_s") if counters else None,
                    "kernel_alone_ratio_to_probe": None if not (ceiling and prof.get("extend_alone_records_per_s"))
                    else round(prof["extend_alone_records_per_s"] / ceiling, 4),
                }
            if counters:
                roof["counters"] = {k: prof.get(k) for k in (
                    "l2_hit_rate", "l1_hit_share", "fabric_share_of_records", "active_lane_fraction",
                    "waves_waiting", "ta_busy", "extend_ms_per_frame_one_stream", "kernel_source_hash", "git_commit")}
            else:
                roof["counters"] = None
            out["roofline"] = roof
        else:
            out["roofline"] = None
        print(json.dumps(out), flush=True)
        fc = out["film_check"]
        if not fc["finite"] or fc["min"] < 0 or (fc["rel_diff"] is not None and fc["rel_diff"] > 0.02) or \
                fc["sample_bit_equal_to_oracle"] is False:
            raise SystemExit("bench: the film of the timed frame fails its check: %r" % (fc,))

    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
